"""HandMapper (src/remap_hand.cpp, remap_hand.hpp): maps a pose of a source hand (ContactGrasp: HumanHand) onto a
destination hand (Allegro) by sampling points along the source's finger chains, scaling them to the destination's chain
lengths and attracting the destination's chain points to them with Point3D3DPriors, under the destination's
self-intersection term, in the NoObs optimiser. The GUI (Pangolin window, sliders) is out of scope; the sliders' role
-- the source hand's articulation -- is a plain array here.

  HandMapper(src, dst, potential)        create_tracker, :59-89 (inter-model intersection zeroed, dst self-intersection on)
  set_src_pose / save_result             the one-line pose files, :92-180
  create_priors                          :388-504 (both hands are assumed to be chains of four frames per finger, as there)
  do_mapping                             :183-202
"""
import math
import os

import numpy as np

from . import se3
from .tracker_no_obs import Point3D3DPrior, TrackerNoObs

F = np.float32


def _length(v):
    return F(np.sqrt(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2])))


def read_pose_line(path):
    """`idx, 12 row-major floats of the palm's model->camera transform, dofs...` (remap_hand.cpp:132-171)."""
    with open(path) as fh:
        vals = [float(t) for t in fh.readline().replace(",", " ").split()]
    return np.asarray(vals[1:13], dtype=F), np.asarray(vals[13:], dtype=F)


def write_pose_line(path, T_cm, dofs):
    """save_result's line: `-1,` + 12 floats + dofs, comma separated (remap_hand.cpp:92-110)."""
    with open(path, "w") as fh:
        fh.write("-1," + ",".join("%g" % float(v) for v in np.asarray(T_cm, dtype=F).reshape(12)))
        fh.write("".join(",%g" % float(v) for v in dofs) + "\n")


def mapped_filename(src_filename, graspit=False):
    if graspit:     # remap_hand.cpp:113-118
        return "graspit_allegro_mapped_" + src_filename[src_filename.find("_") + 1:]
    return "allegro_mapped_" + src_filename


class HandMapper:
    def __init__(self, src_hand, dst_hand, dst_potential, graspit=False, data_dir="grasps"):
        """src_hand / dst_hand: voxelised HostModels; dst_potential: the destination's intersection potential matrix."""
        self.src_hand, self.dst_hand = src_hand, dst_hand
        self.graspit, self.data_dir = graspit, data_dir
        self.src_hand_id, self.dst_hand_id, self.n_priors = 0, 1, 50
        # GUI defaults (remap_hand.cpp:37-41)
        self._max_iterations, self._attract_weight = 160, 150.0
        self._inter_model_intersection, self._intra_model_intersection = 0.0, 5.0
        self._lm_damping, self._log10_regularization = 25.0, -1.0
        self.tracker = TrackerNoObs()
        self.tracker.addModel(src_hand, cloud_seed=31)
        self.tracker.addModel(dst_hand, cloud_seed=32)
        self.tracker.setIntersectionPotentialMatrix(self.dst_hand_id, dst_potential)
        n = self.tracker.getNumModels()
        self.tracker.getOptions().lambdaIntersection = [0.0] * (n * n)
        self.priors = []
        # pose of dst w.r.t. src (:54-55)
        self.T_s_d = se3.mul(se3.mul(se3.translation(-0.14, 0, 0), se3.rot_y(-math.pi / 2)), se3.rot_z(-math.pi / 2)).astype(F)
        self.src_T_cm = se3.identity().astype(F)
        self.src_dofs = np.zeros(src_hand.n_joints, dtype=F)
        self.set_max_iterations(self._max_iterations)
        self.set_intra_model_intersection(self._intra_model_intersection)
        self.set_inter_model_intersection(self._inter_model_intersection)
        self.set_lm_damping(self._lm_damping)
        self.set_log10_regularization(self._log10_regularization)

    # ---- setters (:539-581)
    def set_max_iterations(self, m):
        self._max_iterations = m
        self.tracker.getOptions().numIterations = m

    def set_attract_weight(self, w):
        self._attract_weight = w
        for p in self.priors:
            p.setWeight(w / self.n_priors)

    def set_intra_model_intersection(self, w):
        self._intra_model_intersection = w
        n = self.tracker.getNumModels()
        self.tracker.getOptions().lambdaIntersection[self.dst_hand_id + n * self.dst_hand_id] = w

    def set_inter_model_intersection(self, w):
        self._inter_model_intersection = w
        n = self.tracker.getNumModels()
        self.tracker.getOptions().lambdaIntersection[self.src_hand_id + n * self.dst_hand_id] = w
        self.tracker.getOptions().lambdaIntersection[self.dst_hand_id + n * self.src_hand_id] = w

    def set_log10_regularization(self, w):
        self._log10_regularization = w
        n = self.tracker.getNumModels()
        self.tracker.getOptions().regularization = [float(F(10.0) ** F(w))] * n

    def set_lm_damping(self, w):
        self._lm_damping = w
        self.tracker.getOptions().regularizationScaled = [w] * self.tracker.getNumModels()

    def set_params(self, aw, ad, iw, lmd, reg):
        self.set_attract_weight(aw)
        self.set_inter_model_intersection(iw)
        self.set_intra_model_intersection(iw)
        self.set_lm_damping(lmd)
        self.set_log10_regularization(reg)

    # ---- poses
    def set_src_pose(self, T_cm=None, dofs=None, pose_filename=None):
        if pose_filename is not None:
            T_cm, dofs = read_pose_line(os.path.join(self.data_dir, pose_filename))
            need = self.src_hand.n_joints
            if len(dofs) < need:
                raise ValueError("Need %d dofs in file, got only %d" % (12 + need, 12 + len(dofs)))
            dofs = dofs[:need]
        self.src_T_cm = np.asarray(T_cm, dtype=F).reshape(12)
        lim = np.asarray(self.src_hand.joint_limits, dtype=F).reshape(-1, 2)
        self.src_dofs = np.clip(np.asarray(dofs, dtype=F), lim[:, 0], lim[:, 1]).astype(F)     # the sliders' ranges (:71-78)

    def save_result(self, src_filename):
        path = os.path.join(self.data_dir, mapped_filename(src_filename, self.graspit))
        write_pose_line(path, se3.invert(self.tracker.T_mc[self.dst_hand_id]), self.tracker.joints[self.dst_hand_id])
        return path

    # ---- create_priors (:388-504)
    @staticmethod
    def _chain_lengths(T_mf):
        n_frames = len(T_mf)
        joint = lambda f: T_mf[f][[3, 7, 11]]       # joint_in_model (:381-385): the frame origin in the model   # noqa: E731
        lengths, chain = [], F(0)
        for f in range(1, n_frames):
            if (f - 1) % 4 == 0:
                chain = F(0)
            if (f - 1) % 4 == 3:
                lengths.append(chain)
                continue
            chain = F(chain + _length((joint(f + 1) - joint(f)).astype(F)))
        return lengths

    def create_priors(self):
        tr = self.tracker
        src_T_mf = np.asarray(self.src_hand.forward_kinematics(tr.joints[self.src_hand_id], "dh"), dtype=F).reshape(-1, 12)
        dst_T_mf = np.asarray(self.dst_hand.forward_kinematics(tr.joints[self.dst_hand_id], "dh"), dtype=F).reshape(-1, 12)
        dst_T_cm = se3.invert(tr.T_mc[self.dst_hand_id]).astype(F)
        T_d_s = se3.invert(self.T_s_d).astype(F)
        joint = lambda T, f: T[f][[3, 7, 11]].astype(F)     # noqa: E731
        src_len = self._chain_lengths(src_T_mf)
        src_pts = [[] for _ in src_len]
        src_bases = [np.zeros(3, dtype=F) for _ in src_len]
        finger, dt = -1, F(0)
        for f in range(1, len(src_T_mf)):
            if (f - 1) % 4 == 0:
                finger += 1
                dt = F(src_len[finger] / F(self.n_priors))
            if (f - 1) % 4 == 3:
                continue
            base = joint(src_T_mf, f)
            axis = (joint(src_T_mf, f + 1) - base).astype(F)
            if _length(axis) < 5e-3:
                continue
            unit = (axis / _length(axis)).astype(F)
            t = 0
            while True:
                p = (base + F(F(t) * dt) * unit).astype(F)
                if _length((p - base).astype(F)) > _length(axis):
                    break
                p = (se3.apply(T_d_s, p[None])[0] / src_len[finger]).astype(F)      # into the dst system, unit chain length
                if t == 0 and (f - 1) % 4 == 0:
                    src_bases[finger] = p
                src_pts[finger].append(p)
                t += 1
        dst_len = self._chain_lengths(dst_T_mf)
        dst_pts = [[] for _ in dst_len]
        dst_targs = [[] for _ in dst_len]
        finger, src_finger, src_counter = -1, -1, 0
        base_offset = np.zeros(3, dtype=F)
        for f in range(1, len(dst_T_mf)):
            base = joint(dst_T_mf, f)
            if (f - 1) % 4 == 0:
                finger += 1
                src_finger += 1
                if finger == 3:
                    src_finger += 1         # the destination's fourth finger (thumb) takes the source's fifth chain (:471)
                src_counter = 0
                dt = F(dst_len[finger] / F(self.n_priors))
                base_offset = (base - src_bases[src_finger] * dst_len[finger]).astype(F)
            if (f - 1) % 4 == 3:
                continue
            axis = (joint(dst_T_mf, f + 1) - base).astype(F)
            if _length(axis) < 5e-3:
                continue
            unit = (axis / _length(axis)).astype(F)
            T_fm = se3.invert(dst_T_mf[f]).astype(F)
            t = 0
            while True:
                p = (base + F(F(t) * dt) * unit).astype(F)
                if _length((p - base).astype(F)) > _length(axis):
                    break
                if src_counter == len(src_pts[src_finger]):
                    break
                dst_pts[finger].append((se3.apply(T_fm, p[None])[0].astype(F), f))
                src_p = (src_pts[src_finger][src_counter] * dst_len[finger] + base_offset).astype(F)
                dst_targs[finger].append(se3.apply(dst_T_cm, src_p[None])[0].astype(F))
                t += 1
                src_counter += 1
        tr.clearPriors()
        self.priors = []
        for finger in range(len(dst_pts)):
            for (pf, f), targ in zip(dst_pts[finger], dst_targs[finger]):
                pr = Point3D3DPrior(self.dst_hand_id, f, tuple(float(v) for v in targ), tuple(float(v) for v in pf), self._attract_weight)
                self.priors.append(pr)
                tr.addPrior(pr)
        return self.priors

    # ---- do_mapping (:183-202)
    def do_mapping(self):
        tr = self.tracker
        tr.setPose(self.src_hand_id, se3.invert(self.src_T_cm), self.src_dofs)
        dst_T_cm = se3.mul(self.src_T_cm, self.T_s_d).astype(F)
        tr.setPose(self.dst_hand_id, se3.invert(dst_T_cm), np.zeros(self.dst_hand.n_joints, dtype=F))
        self.create_priors()
        return tr.optimizePoses()
