"""GraspAnalyser (src/grasp_analyzer.{hpp,cpp}) without the GUI: ContactGrasp's file formats and its batch driver.

The reference pushes the grasps one by one through a queue served by the Pangolin thread (grasp_analyzer.cpp:851-902,
:714-760), each a 40-iteration TrackerNoObs::optimizePoses(). Here the whole list is one batch of independent candidates
(TrackerNoObs.optimizePosesBatch), sharded round-robin over the ranks when torch.distributed is initialised, and gathered
with one all-gather of {energy, pose}.

File formats (all text):
  <obj>_grasps[_allegro|_barrett].csv     first line a comment; then per grasp 12 floats = the palm pose, rows of the 3x4
                                          model->camera transform, followed by the DOFs (grasp_analyzer.cpp:117-151, :262-281)
  <session>_<object>_contact_info.txt     per line `is_contact px py pz nx ny nz`, points multiplied by scale/1.15 (:316-345)
  [prefix]<session>_<object>_gt_hand_pose.txt   `idx,` + a grasp line (:299-314)
  [prefix]<session>_<object>_grasp_errors.csv   one line per analysed grasp: comma separated energies (:904-928)
  thumb.txt                               `frame px py pz` (:226-235)
  intersection_potential_matrix.txt       nSdfs x nSdfs 0/1 matrix (geometry.cpp loadSelfIntersectionMatrix)
"""
import os

import numpy as np

from . import configs, se3, sharding
from .tracker_no_obs import ContactPrior, TrackerNoObs

F = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# std::shuffle with a default-constructed std::default_random_engine, as libstdc++ implements it (bits/stl_algo.h shuffle,
# bits/uniform_int_dist.h): the reference picks its 500 contact / 2000 non-contact points with it, one engine for both
# lists (grasp_analyzer.cpp:346-365). Pinned by tests/golden/std_shuffle.json (generated with g++'s own library).
# ---------------------------------------------------------------------------------------------------------------------
class MinStdRand0:
    """std::minstd_rand0 = linear_congruential_engine<uint_fast32_t, 16807, 0, 2147483647>, default seed 1."""
    MIN, MAX = 1, 2147483646

    def __init__(self, seed=1):
        self.x = seed % 2147483647 or 1

    def __call__(self):
        self.x = (self.x * 16807) % 2147483647
        return self.x


def _uniform_int(g, a, b):
    """uniform_int_distribution<uint64>{a, b}(g) for a generator narrower than 32 or 64 full bits (the 2-division path)."""
    urngrange = g.MAX - g.MIN
    urange = b - a
    if urngrange > urange:
        uerange = urange + 1
        scaling = urngrange // uerange
        past = uerange * scaling
        while True:
            ret = g() - g.MIN
            if ret < past:
                break
        ret //= scaling
    elif urngrange < urange:
        uerngrange = urngrange + 1
        while True:
            tmp = uerngrange * _uniform_int(g, 0, urange // uerngrange)
            ret = tmp + (g() - g.MIN)
            if ret <= urange:
                break
    else:
        ret = g() - g.MIN
    return ret + a


def std_shuffle(seq, g):
    """In-place std::shuffle(seq.begin(), seq.end(), g) of libstdc++ (pairs of swaps from one draw when the range allows)."""
    n = len(seq)
    if n == 0:
        return seq
    urngrange = g.MAX - g.MIN
    if urngrange // n >= n:
        i = 1
        if n % 2 == 0:
            j = _uniform_int(g, 0, 1)
            seq[i], seq[j] = seq[j], seq[i]
            i += 1
        while i != n:
            swap_range = i + 1
            x = _uniform_int(g, 0, swap_range * (swap_range + 1) - 1)
            p0, p1 = x // (swap_range + 1), x % (swap_range + 1)
            seq[i], seq[p0] = seq[p0], seq[i]
            i += 1
            seq[i], seq[p1] = seq[p1], seq[i]
            i += 1
        return seq
    for i in range(1, n):
        j = _uniform_int(g, 0, i)
        seq[i], seq[j] = seq[j], seq[i]
    return seq


# ---------------------------------------------------------------------------------------------------------------------
# file formats
# ---------------------------------------------------------------------------------------------------------------------
def pose_from_string(line, n_dofs, dof_mapping=None, read_gt=False):
    """pose_from_string (grasp_analyzer.cpp:117-151): returns (T_cm [12] model->camera, dofs [n_dofs], gt index or None);
    None when the line holds fewer than 12 + n_dofs values (the reference prints a warning and leaves the pose alone)."""
    vals = [float(v) for v in line.replace(",", " ").split()]
    gt = None
    if read_gt:
        gt = int(vals[0])
        vals = vals[1:]
    if len(vals) < 12 + n_dofs:
        return None
    T_cm = np.asarray(vals[:12], dtype=F)
    dofs = np.zeros(n_dofs, dtype=F)
    mapping = range(n_dofs) if dof_mapping is None else dof_mapping
    for i in range(n_dofs):
        dofs[mapping[i]] = vals[12 + i]
    return T_cm, dofs, gt


def _fmt(v):
    return "%g" % float(v)          # operator<< of a float: 6 significant digits


def string_from_pose(T_cm, dofs):
    """string_from_pose (grasp_analyzer.cpp:154-173)."""
    return ",".join(_fmt(v) for v in list(np.asarray(T_cm).reshape(12)) + list(np.asarray(dofs).reshape(-1)))


def read_grasps_csv(path, n_dofs, dof_mapping=None):
    """The grasp list (grasp_analyzer.cpp:262-281): first line is a comment. Returns T_cm [G,12], dofs [G,n_dofs]."""
    Ts, Js = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            if not line.strip():
                continue
            r = pose_from_string(line, n_dofs, dof_mapping)
            if r is None:       # the reference pushes the unmodified template pose; keep the list aligned the same way
                Ts.append(se3.identity().astype(F))
                Js.append(np.zeros(n_dofs, dtype=F))
            else:
                Ts.append(r[0])
                Js.append(r[1])
    if not Ts:
        return np.zeros((0, 12), dtype=F), np.zeros((0, n_dofs), dtype=F)
    return np.stack(Ts), np.stack(Js)


def write_grasps_csv(path, T_cms, dofs, comment="# palm pose (3x4 row-major), dofs"):
    with open(path, "w") as f:
        f.write(comment + "\n")
        for T, j in zip(T_cms, dofs):
            f.write(string_from_pose(T, j) + "\n")


def read_contact_info(path, scale=1.0):
    """`is_contact px py pz nx ny nz` per line; points scaled by scale/1.15 (grasp_analyzer.cpp:316-345).
    Returns (contact points, contact normals, no-contact points, no-contact normals)."""
    s = F(scale) / F(1.15)
    cp, cn, np_, nn = [], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 7:
                continue
            p = [F(float(t[1])) * s, F(float(t[2])) * s, F(float(t[3])) * s]
            n = [F(float(t[4])), F(float(t[5])), F(float(t[6]))]
            if int(float(t[0])):
                cp.append(p); cn.append(n)
            else:
                np_.append(p); nn.append(n)

    def arr(x):
        return np.asarray(x, dtype=F).reshape(-1, 3)
    return arr(cp), arr(cn), arr(np_), arr(nn)


def read_thumb_file(path):
    """`frame px py pz` (grasp_analyzer.cpp:226-235)."""
    with open(path) as f:
        t = f.read().split()
    return int(t[0]), (float(t[1]), float(t[2]), float(t[3]))


def read_potential_matrix(path, n_sdfs):
    with open(path) as f:
        v = [int(float(t)) for t in f.read().split()]
    if len(v) < n_sdfs * n_sdfs:
        raise ValueError("%s holds %d entries, need %d" % (path, len(v), n_sdfs * n_sdfs))
    return np.asarray(v[:n_sdfs * n_sdfs], dtype=np.int32)


def write_grasp_errors(path, results):
    """save_results (grasp_analyzer.cpp:904-928): one line per analysed grasp, values joined by commas."""
    with open(path, "w") as f:
        for r in results:
            f.write(",".join(_fmt(v) for v in r) + "\n")


def read_grasp_errors(path, col, n_grasps):
    """The column set_order sorts by (grasp_analyzer.cpp:814-849); missing lines stay at FLT_MAX."""
    errors = np.full(n_grasps, np.finfo(F).max, dtype=F)
    with open(path) as f:
        for i, line in enumerate(f):
            if i >= n_grasps:
                break
            errors[i] = float(line.split(",")[col])
    return errors


def compare_poses(frames_a, frames_b):
    """compare_poses (grasp_analyzer.cpp:175-183): sum of squared frame-origin distances."""
    a, b = np.asarray(frames_a, dtype=F).reshape(-1, 12), np.asarray(frames_b, dtype=F).reshape(-1, 12)
    d = a[:, [3, 7, 11]] - b[:, [3, 7, 11]]
    return float((d.astype(F) ** 2).sum(dtype=F))


# ---------------------------------------------------------------------------------------------------------------------
# the analyser
# ---------------------------------------------------------------------------------------------------------------------
class GraspAnalyser:
    """GraspAnalyser without the Pangolin thread. `hand` / `obj` are voxelised HostModels (the reference loads XML + meshes
    by name; mesh import is outside this path), `hand_name` picks the file-name prefixes as the reference does."""

    def __init__(self, object_name, session_name, hand, obj, data_dir, hand_name="human_hand", graspit=False, scale=1.0,
                 potential=None, thumb=None, max_batch=4096):
        self.object_name, self.session_name, self.data_dir = object_name, session_name, data_dir
        self.hand_name, self.graspit, self.scale = hand_name, graspit, scale
        # GUI defaults (grasp_analyzer.cpp:198-204)
        self._max_iterations = 40
        self._attract_weight, self._repulse_weight, self._thumb_attract_weight = 150.0, 20.0, 25.0
        self._attract_dist_cm, self._repulse_dist_cm = 2.0, 2.0
        self._inter_model_intersection, self._intra_model_intersection = 100.0, 5.0
        self._lm_damping, self._log10_regularization = 25.0, -7.5
        self.hand_id, self.object_id = 0, 1
        # create_tracker (:392-436)
        self.tracker = TrackerNoObs(max_batch=max_batch)
        # the Barrett hand couples its 8 joints into 4 (LinearPoseReduction(8, 4), :393-420); every other hand is unreduced
        self.reduction = configs.barrett_reduction(hand) if hand_name == "barrett" else None
        self.tracker.addModel(hand, cloud_seed=11, reduction=self.reduction)
        if potential is not None:
            self.tracker.setIntersectionPotentialMatrix(self.hand_id, potential)
        self.hand, self.obj = hand, obj
        # sized by getReducedArticulatedDimensions(), identity mapping (:427-435)
        self.n_dofs = hand.n_joints if self.reduction is None else self.reduction.getReducedDimensions()
        self.graspit2dartdofs = list(range(self.n_dofs))
        self.thumb_frame_id, self.thumb_contact_point = thumb if thumb is not None else (0, (0.0, 0.0, 0.0))
        self.grasps_T_cm = np.zeros((0, 12), dtype=F)
        self.grasps_dofs = np.zeros((0, self.n_dofs), dtype=F)
        self.order = []
        self.contact_priors = []
        self.thumb_contact_prior = None
        self.gt_frame_poses = None
        self.results = []
        self.optimized = None

    def _full_joints(self, dofs):
        """Pose::projectReducedToFull for the hand: the full joint vector of a (possibly reduced) articulation."""
        if self.reduction is None:
            return np.asarray(dofs, dtype=F)
        full = np.zeros(self.hand.n_joints, dtype=F)
        self.reduction.projectReducedToFull(np.asarray(dofs, dtype=F), full)
        return full

    # ---- file names (:252-296, :814-820, :904-911)
    def _obj_base(self):
        return self.object_name.split("-")[0]

    def _hand_prefix(self, with_mapped=True):
        if self.hand_name == "allegro_mapped":
            return "allegro_mapped_" if with_mapped else "allegro_"
        if self.hand_name == "allegro":
            return "allegro_"
        if self.hand_name == "barrett":
            return "barrett_"
        return ""

    def grasps_filename(self):
        suffix = {"allegro": "_grasps_allegro.csv", "allegro_mapped": "_grasps_allegro.csv", "barrett": "_grasps_barrett.csv"}
        return os.path.join(self.data_dir, self._obj_base() + suffix.get(self.hand_name, "_grasps.csv"))

    def gt_filename(self):
        return os.path.join(self.data_dir, ("graspit_" if self.graspit else "") + self._hand_prefix() +
                            "%s_%s_gt_hand_pose.txt" % (self.session_name, self.object_name))

    def contact_filename(self):
        return os.path.join(self.data_dir, "%s_%s_contact_info.txt" % (self.session_name, self.object_name))

    def errors_filename(self, for_reading=False):
        # save_results prefixes "graspit_" and "allegro_mapped_", set_order neither (:814-820 vs :904-911)
        pre = self._hand_prefix(with_mapped=False) if for_reading else ("graspit_" if self.graspit else "") + self._hand_prefix()
        return os.path.join(self.data_dir, pre + "%s_%s_grasp_errors.csv" % (self.session_name, self.object_name))

    # ---- load_object (:248-389)
    def load_object(self):
        t = self.tracker
        t.addModel(self.obj, cloud_seed=12)
        self.grasps_T_cm, self.grasps_dofs = read_grasps_csv(self.grasps_filename(), self.n_dofs, self.graspit2dartdofs)
        if len(self.grasps_T_cm) == 0:
            return
        self.reset_order()
        if os.path.exists(self.gt_filename()):
            with open(self.gt_filename()) as f:
                r = pose_from_string(f.readline(), self.n_dofs, self.graspit2dartdofs, read_gt=True)
            if r is not None:       # frame -> camera transforms of the ground-truth grasp (the object pose is identity)
                T_mf = np.asarray(self.hand.forward_kinematics(self._full_joints(r[1]), "dh"), dtype=F).reshape(-1, 12)
                self.gt_frame_poses = np.stack([se3.mul(r[0], T) for T in T_mf])
        cp, cn, ncp, ncn = read_contact_info(self.contact_filename(), self.scale)
        self.n_attract_points = min(len(cp), 500)
        rng = MinStdRand0()
        idx = std_shuffle(list(range(len(cp))), rng)[:self.n_attract_points]
        n_div = max(self.n_attract_points, 1)
        for i in idx:
            p = ContactPrior(self.object_id, self.hand_id, 0, -1, self._attract_weight / n_div, tuple(cp[i]), False, True, True, False,
                             self._attract_dist_cm / 100.0, tuple(cn[i]))
            t.addPrior(p)
            self.contact_priors.append(p)
        self.n_repulse_points = min(len(ncp), 2000)
        idx = std_shuffle(list(range(len(ncp))), rng)[:self.n_repulse_points]
        for i in idx:
            p = ContactPrior(self.object_id, self.hand_id, 0, -1, self._repulse_weight / n_div, tuple(ncp[i]), False, True, True, True,
                             self._repulse_dist_cm / 100.0, tuple(ncn[i]))
            t.addPrior(p)
            self.contact_priors.append(p)
        self.thumb_contact_prior = ContactPrior(self.hand_id, self.object_id, self.thumb_frame_id, 0, self._thumb_attract_weight,
                                                tuple(self.thumb_contact_point), False, True, True, False, self._attract_dist_cm / 100.0)
        t.addPrior(self.thumb_contact_prior)
        n = t.getNumModels()
        t.getOptions().lambdaIntersection = [0.0] * (n * n)
        self.apply_gui_defaults()

    def apply_gui_defaults(self):
        """The slider callbacks fire once with their defaults when the GUI comes up (:459-476)."""
        self.set_max_iterations(self._max_iterations)
        self.set_attract_weight(self._attract_weight)
        self.set_repulse_weight(self._repulse_weight)
        self.set_thumb_attract_weight(self._thumb_attract_weight)
        self.set_attract_dist_cm(self._attract_dist_cm)
        self.set_repulse_dist_cm(self._repulse_dist_cm)
        self.set_intra_model_intersection(self._intra_model_intersection)
        self.set_inter_model_intersection(self._inter_model_intersection)
        self.set_log10_regularization(self._log10_regularization)
        self.set_lm_damping(self._lm_damping)

    # ---- setters (:936-1008)
    def set_max_iterations(self, m):
        self._max_iterations = m
        self.tracker.getOptions().numIterations = m

    def set_attract_weight(self, w):
        self._attract_weight = w
        for p in self.contact_priors:
            if not p.isInverted():
                p.setWeight(w)

    def set_repulse_weight(self, w):
        self._repulse_weight = w
        for p in self.contact_priors:
            if p.isInverted():
                p.setWeight(w)

    def set_thumb_attract_weight(self, w):
        self._thumb_attract_weight = w
        self.thumb_contact_prior.setWeight(w)

    def set_attract_dist_cm(self, d):
        self._attract_dist_cm = d
        for p in self.contact_priors:
            if not p.isInverted():
                p.setContactThreshold(d / 100.0)
        self.thumb_contact_prior.setContactThreshold(d / 100.0)

    def set_repulse_dist_cm(self, d):
        self._repulse_dist_cm = d
        for p in self.contact_priors:
            if p.isInverted():
                p.setContactThreshold(d / 100.0)

    def set_intra_model_intersection(self, w):
        self._intra_model_intersection = w
        n = self.tracker.getNumModels()
        self.tracker.getOptions().lambdaIntersection[self.hand_id + n * self.hand_id] = w

    def set_inter_model_intersection(self, w):
        self._inter_model_intersection = w
        n = self.tracker.getNumModels()
        lam = self.tracker.getOptions().lambdaIntersection
        lam[self.object_id + n * self.hand_id] = w
        lam[self.hand_id + n * self.object_id] = w

    def set_log10_regularization(self, w):
        self._log10_regularization = w
        v = float(F(10.0) ** F(w))
        o = self.tracker.getOptions()
        o.regularization = [v] * len(o.regularization)

    def set_lm_damping(self, w):
        self._lm_damping = w
        o = self.tracker.getOptions()
        o.regularizationScaled = [w] * len(o.regularizationScaled)

    # ---- ordering (:762-849, :930-934)
    def reset_order(self):
        self.order = list(range(len(self.grasps_T_cm)))

    def set_order(self, col_idx):
        path = self.errors_filename(for_reading=True)
        if not os.path.exists(path):
            return
        errors = read_grasp_errors(path, col_idx, len(self.grasps_T_cm))
        self.order = sorted(self.order, key=lambda i: errors[i])     # std::sort is not stable; ties may order differently

    # ---- analyze_grasps (:851-902) + the queue's body (:714-760) + save_results
    def analyze_grasps(self, n=-1, order_by=-1, energy_type=0, save=True):
        """Refines the first n grasps of the current order as one batch (sharded over the ranks of torch.distributed when
        initialised) and returns the mean of the requested energy (0: DART energy, 1: GT similarity, -1: both columns)."""
        import torch
        import torch.distributed as dist
        if order_by >= 0:
            self.set_order(order_by)
        else:
            self.reset_order()
        if n < 0:
            n = len(self.grasps_T_cm)
        todo = self.order[:n]
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        mine = sharding.shard_indices(len(todo), rank, world)
        nj = self.n_dofs
        T = np.zeros((len(mine), 2, 12), dtype=F)
        for k, q in enumerate(mine):
            T[k, 0] = se3.invert(self.grasps_T_cm[todo[q]])       # the file holds model->camera, the tracker camera->model
            T[k, 1] = se3.identity()
        j = np.stack([self.grasps_dofs[todo[q]] for q in mine]) if len(mine) else np.zeros((0, nj), dtype=F)
        j = np.concatenate([j, np.zeros((len(mine), self.obj.n_joints), dtype=F)], axis=1) if self.obj.n_joints else j
        dev = torch.device("cuda", torch.cuda.current_device())
        if len(mine):
            T_o, j_o, e_o, _, _ = self.tracker.optimizePosesBatch(torch.from_numpy(T).to(dev), torch.from_numpy(j).to(dev), device_arrays=True)
        else:
            T_o = torch.zeros((0, 2, 12), device=dev); j_o = torch.zeros((0, j.shape[1]), device=dev); e_o = torch.zeros(0, device=dev)
        poses = torch.cat([T_o.reshape(len(mine), -1), j_o], dim=1)
        energy, poses = sharding.gather_candidates(len(todo), rank, world, e_o, poses)
        energy, poses = energy.cpu().numpy(), poses.cpu().numpy()
        self.optimized = poses
        self.results = []
        for q in range(len(todo)):
            r = []
            if energy_type <= 0:
                r.append(float(energy[q]))
            # the reference tests `type == 2 || type < 0` (:735): energy type 1 alone never yields the GT column there
            if (energy_type == 2 or energy_type < 0) and self.gt_frame_poses is not None:
                T_mc_h, T_mc_o = poses[q, 0:12], poses[q, 12:24]
                T_mf = np.asarray(self.hand.forward_kinematics(self._full_joints(poses[q, 24:24 + nj]), "dh"), dtype=F).reshape(-1, 12)
                T_cm_h = se3.invert(T_mc_h)
                frames = [se3.mul(T_mc_o, se3.mul(T_cm_h, Tf)) for Tf in T_mf[:len(self.gt_frame_poses)]]
                r.append(compare_poses(frames, self.gt_frame_poses))
            self.results.append(r)
        if save and rank == 0:
            write_grasp_errors(self.errors_filename(), self.results)
        col = max(0, energy_type)
        vals = [r[col] for r in self.results if len(r) > col]
        return float(np.mean(vals)) if vals else float("nan")
