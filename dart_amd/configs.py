"""The synthetic workloads of BASELINE.json `configs`: every input is generated from seed 0xDA27.

C1  rigid sphere r = 0.10 m at 0.6 m, one 80^3 SDF, 640x480 depth by SDF ray-cast.
C2  Allegro hand (kinematics from models/allegro/allegro.xml; declared primitive proxies for the missing meshes),
    640x480 depth of the TRUE pose, estimate starts 1.5 cm / 4 deg / 0.15 rad per joint away, prediction 320x240,
    64^3 observation SDF, lambda_obsToMod = lambda_modToObs = 1.
"""
from dataclasses import dataclass

import numpy as np

from . import se3
from .model import load_asset, sphere_model
from .optimizer import OptimizationOptions

F = np.float32
SEED = 0xDA27
FOCAL = 537.0     # optimization.h:43
# The single-model association rejects a point when dot(-grad_c, n_obs) < normThreshold (obsToMod.cu:84-96): with the
# reference's own normal maps (outward normals, organized_point_cloud.cu:286-299) and the default -0.1 that gate is
# inverted and drops ~94 % of the well-aligned points. The workloads disable the gate
# (threshold below -1) so that EVERY associated point enters the normal equations -- the heavier, conservative
# choice for a benchmark; tests cover the default -0.1 as well.
NORM_GATE_OFF = -2.0


@dataclass
class TrackingScene:
    name: str
    host_model: object
    obs_layout: tuple
    width: int
    height: int
    fl: tuple
    pp: tuple
    true_T_mc: np.ndarray
    true_joints: np.ndarray
    init_T_mc: np.ndarray
    init_joints: np.ndarray
    opts: OptimizationOptions
    pred_width: int
    pred_height: int


def _perturb(T_mc, joints, limits, rng, trans, rot_deg, joint_delta):
    d = rng.normal(size=3)
    d = d / np.linalg.norm(d) * trans
    a = rng.normal(size=3)
    a = a / np.linalg.norm(a) * np.deg2rad(rot_deg)
    T = se3.mul(se3.exp(np.concatenate([d, a]).astype(F)), T_mc)
    j = joints + rng.choice([-1.0, 1.0], size=len(joints)) * joint_delta if len(joints) else joints
    if len(joints):
        j = np.clip(j, limits[:, 0], limits[:, 1])
    return T, np.asarray(j, dtype=F)


def config1(width=640, height=480):
    rng = np.random.default_rng(SEED)
    m = sphere_model(0.1).voxelize2(0.005, 0.10)
    T_cm = se3.translation(0, 0, 0.6)
    T_mc = se3.invert(T_cm)
    init_T, init_j = _perturb(T_mc, np.zeros(0, dtype=F), None, rng, 0.02, 5.0, 0.0)
    opts = OptimizationOptions(numIterations=5)
    opts.distThreshold = [0.03]
    opts.normThreshold = NORM_GATE_OFF
    s = width / 640.0
    return TrackingScene("C1-sphere", m, m.obs_sdf_layout(), width, height, (FOCAL * s, FOCAL * s), (width / 2, height / 2),
                         T_mc, np.zeros(0, dtype=F), init_T, init_j, opts, width // 2, height // 2)


def _articulated(name, asset, width, height, z, rng, trans=0.015, rot=4.0, jd=0.15, pose_frac=(0.25, 0.6)):
    m = load_asset(asset).voxelize(0.005, 0.10)
    lim = np.asarray(m.joint_limits, dtype=np.float64).reshape(-1, 2)
    joints = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(pose_frac[0], pose_frac[1], len(lim))).astype(F)
    # palm facing the camera, wrist down: rotate the model frame so that the fingers (model +z for allegro) point up (-y)
    T_cm = se3.mul(se3.translation(0.0, 0.03, z), se3.exp(np.array([0, 0, 0, 1.2, 0.2, -0.1], dtype=F)))
    T_mc = se3.invert(T_cm)
    init_T, init_j = _perturb(T_mc, joints.astype(np.float64), lim, rng, trans, rot, jd)
    opts = OptimizationOptions(numIterations=5)
    opts.normThreshold = NORM_GATE_OFF
    s = width / 640.0
    return TrackingScene(name, m, m.obs_sdf_layout(), width, height, (FOCAL * s, FOCAL * s), (width / 2, height / 2),
                         T_mc, joints, init_T, init_j, opts, width // 2, height // 2)


def config2(width=640, height=480):
    return _articulated("C2-allegro", "allegro", width, height, 0.45, np.random.default_rng(SEED))


def tracking_frames(sc, n, seed=SEED + 100):
    """n independent frames of a single-model tracking scene (a recorded sequence's stand-in for Optimizer::optimizePose looped over
    frames): frame 0 is the scene itself, frame f > 0 a seeded perturbation of its true pose with its own perturbed start pose.
    Returns (true_T [n, 12], true_joints [n, nJ], init_T [n, 12], init_joints [n, nJ])."""
    m = sc.host_model
    nj = len(sc.true_joints)
    lim = np.asarray(m.joint_limits, dtype=np.float64).reshape(-1, 2) if nj else None
    tT, tj, iT, ij = [sc.true_T_mc], [sc.true_joints], [sc.init_T_mc], [sc.init_joints]
    for f in range(1, n):
        rng = np.random.default_rng(seed + f)
        T, j = _perturb(sc.true_T_mc, sc.true_joints.astype(np.float64), lim, rng, 0.02 * rng.uniform(0.2, 1.0), 6.0 * rng.uniform(0.2, 1.0), 0.1)
        T0, j0 = _perturb(T, j.astype(np.float64), lim, rng, 0.012, 3.0, 0.1)
        tT.append(T); tj.append(j); iT.append(T0); ij.append(j0)
    return (np.stack(tT).astype(F), np.stack(tj).astype(F).reshape(n, nj), np.stack(iT).astype(F), np.stack(ij).astype(F).reshape(n, nj))


def config2_left_hand(width=640, height=480):
    """The asset-faithful articulated model (primitives straight from models/leftHand/leftHand.xml)."""
    return _articulated("C2-leftHand", "left_hand", width, height, 0.45, np.random.default_rng(SEED + 1))


# ---------------------------------------------------------------------------------------------------------------------
# ContactGrasp (BASELINE configs 3 / 4): HumanHand + object, attractive / repulsive contact points + thumb prior.
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class GraspScene:
    name: str
    hand: object                 # voxelised HostModel (model 0 = hand_id, grasp_analyzer.cpp:204)
    obj: object                  # voxelised HostModel (model 1 = object_id)
    potential: np.ndarray        # hand self-intersection potential matrix [nSdfs*nSdfs]
    priors: list                 # tracker_no_obs.ContactPrior
    gt_T_mc: np.ndarray          # "ground truth" hand pose the contact map was painted from
    gt_joints: np.ndarray
    init_T_mc: np.ndarray        # [B, 12] candidate hand poses
    init_joints: np.ndarray      # [B, nJ]
    num_iterations: int
    lambda_intra: float
    lambda_inter: float
    lm_damping: float
    regularization: float


OBJ_RADII = (0.045, 0.035, 0.06)    # ellipsoid "object" (the reference's object meshes are not in its tree)


def _object_model():
    from .model import HostModel, GEOM_SPHERE
    m = HostModel(version=1, name="object")
    m._geom_specs = [{"frame": 0, "type": GEOM_SPHERE, "scale": list(OBJ_RADII), "trans": [0, 0, 0], "rot": [0, 0, 0]}]
    m.add_geometry(0, GEOM_SPHERE, list(OBJ_RADII), [0, 0, 0], [0, 0, 0])
    return m.voxelize(0.005, 0.10)


def _hand_potential(asset, n):
    import json
    import os
    from .model import ASSET_DIR
    spec = json.load(open(os.path.join(ASSET_DIR, asset + ".json")))
    pot = np.zeros((n, n), dtype=np.int32)
    mat = np.asarray(spec["intersection_potential"], dtype=np.int32)
    k = min(n, mat.shape[0])
    pot[:k, :k] = mat[:k, :k]
    return pot.reshape(-1), spec.get("thumb")


def contact_grasp_scene(n_candidates=1, asset="human_hand", n_attract=500, n_repulse=2000, iterations=40, seed=SEED):
    """GraspAnalyser's cost set-up (grasp_analyzer.cpp:198-204, 248-389, 941-1008) on synthetic contact data: the object
    sits at the origin of the camera frame, the contact map is "painted" by a ground-truth grasp (object surface points
    within 1 cm of the posed hand are contact points, the rest non-contact points, outward normals), the candidates are
    seeded perturbations of that grasp."""
    from scipy.spatial import cKDTree
    from .tracker_no_obs import ContactPrior
    rng = np.random.default_rng(seed)
    hand = load_asset(asset).voxelize(0.005, 0.10)
    obj = _object_model()
    pot, thumb = _hand_potential(asset, len(hand.sdfs))
    lim = np.asarray(hand.joint_limits, dtype=np.float64).reshape(-1, 2)
    # palm below the object, fingers curling around it: for a few finger-flexion levels slide the palm along the
    # approach axis until the hand's collision cloud just touches the ellipsoid, and keep the grasp whose links touch most
    cloud = hand.collision_cloud(1e5, seed=1)
    a = np.asarray(OBJ_RADII, dtype=np.float64)
    best = (-1, None, None)
    for flex in (0.25, 0.35, 0.45, 0.55, 0.65):
        jt = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * flex).astype(F)
        T_mf = hand.forward_kinematics(jt, "dh")
        sdf_of = cloud[:, 3].astype(int)
        pts_m = np.zeros((len(cloud), 3), dtype=F)
        for sn in np.unique(sdf_of):
            sel = sdf_of == sn
            pts_m[sel] = se3.apply(T_mf[hand.sdf_frame[int(sn)]], cloud[sel, :3])
        prev = None
        # HumanHand: palm at x ~ -0.08, fingers along -x, flexion curls them towards -y: the object approaches from -y
        for dist in np.arange(0.20, 0.0, -0.002):
            T_try = se3.translation(0.125, dist, 0.0)       # model -> camera (object frame)
            pc = se3.apply(T_try, pts_m).astype(np.float64)
            lvl = np.sqrt(((pc / a) ** 2).sum(1))
            if (lvl < 1.0).sum() > 20:
                break
            prev = (T_try, pc, lvl)
        if prev is None:
            continue
        T_try, pc, lvl = prev
        near = np.abs(lvl - 1.0) * a.min() < 0.008
        score = len(np.unique(sdf_of[near]))
        if score > best[0]:
            best = (score, T_try, jt, pts_m)
    _, T_cm, gt_joints, pts_m = best
    gt_T_mc = se3.invert(T_cm)
    # object surface samples + outward normals, split by distance to the posed hand
    n_s = 20000
    v = rng.normal(size=(n_s, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    surf = v * a
    nrm = surf / a ** 2
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hand_c = se3.apply(T_cm, pts_m.astype(F)).astype(np.float64)
    d, _ = cKDTree(hand_c).query(surf)
    contact = np.where(d < 0.012)[0]
    free = np.where(d >= 0.012)[0]
    rng.shuffle(contact)
    rng.shuffle(free)
    contact, free = contact[:n_attract], free[:n_repulse]
    # the order of the priors only fixes the order of the sums; keep neighbours together (spatial.py)
    from .spatial import morton_order
    contact = contact[morton_order(surf[contact])]
    free = free[morton_order(surf[free])]
    att_w, rep_w, thumb_w = 150.0, 20.0, 25.0      # after the GUI setters ran (undivided, grasp_analyzer.cpp:941-965)
    thr = 0.02
    priors = []
    for i in contact:      # ContactPrior(object_id, hand_id, 0, -1, w, p, false, true, true, false, thr, n)
        priors.append(ContactPrior(1, 0, 0, -1, att_w, tuple(surf[i].astype(F)), False, True, True, False, thr, tuple(nrm[i].astype(F))))
    for i in free:
        priors.append(ContactPrior(1, 0, 0, -1, rep_w, tuple(surf[i].astype(F)), False, True, True, True, thr, tuple(nrm[i].astype(F))))
    if thumb is not None:  # ContactPrior(hand_id, object_id, thumb_frame_id, 0, w, thumb_point, false, true, true, false, thr)
        priors.append(ContactPrior(0, 1, int(thumb["sdf"]), 0, thumb_w, tuple(thumb["point"]), False, True, True, False, thr))
    # candidates: candidate 0 = a mild perturbation (C3), the rest +-3 cm / +-20 deg / joints uniform in limits (C4)
    Ts, Js = [], []
    for c in range(n_candidates):
        if c == 0:
            T, j = _perturb(gt_T_mc, gt_joints.astype(np.float64), lim, rng, 0.01, 5.0, 0.1)
        else:
            T, _ = _perturb(gt_T_mc, gt_joints.astype(np.float64), lim, rng, 0.03 * rng.random(), 20.0 * rng.random(), 0.0)
            j = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.random(len(lim))).astype(F)
        Ts.append(T)
        Js.append(j)
    return GraspScene("C3-contactgrasp" if n_candidates == 1 else "C4-contactgrasp-batch", hand, obj, pot, priors, gt_T_mc, gt_joints,
                      np.stack(Ts).astype(F), np.stack(Js).astype(F), iterations, 5.0, 100.0, 25.0, float(F(10.0) ** F(-7.5)))


def make_grasp_tracker(sc, max_batch=1):
    """create_tracker() + load_object() + the option setters (grasp_analyzer.cpp:392-436, 248-389, 936-1008)."""
    from .tracker_no_obs import TrackerNoObs
    tr = TrackerNoObs(max_batch=max_batch)
    hand_id = tr.addModel(sc.hand, cloud_seed=11)
    tr.setIntersectionPotentialMatrix(hand_id, sc.potential)
    obj_id = tr.addModel(sc.obj, cloud_seed=12)
    for p in sc.priors:
        tr.addPrior(p)
    n = tr.getNumModels()
    o = tr.getOptions()
    o.numIterations = sc.num_iterations
    o.lambdaIntersection[hand_id + n * hand_id] = sc.lambda_intra
    o.lambdaIntersection[obj_id + n * hand_id] = sc.lambda_inter
    o.lambdaIntersection[hand_id + n * obj_id] = sc.lambda_inter
    o.regularization = [sc.regularization] * n
    o.regularizationScaled = [sc.lm_damping] * n
    return tr


def grasp_batch_arrays(sc):
    """[B, 2, 12] camera->model transforms (hand, object = identity) and [B, nJ] joints of the candidates."""
    B = len(sc.init_T_mc)
    T = np.zeros((B, 2, 12), dtype=F)
    T[:, 0] = sc.init_T_mc
    T[:, 1] = se3.identity()
    return T, sc.init_joints.copy()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: Barrett hand (8 joints reduced to 4, grasp_analyzer.cpp:393-420) + object, multi-model tracking with
# observations, self-intersection 5 and hand<->object intersection 100, 1280x720 depth.
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class MultiScene:
    name: str
    hand: object
    obj: object
    reduction: object
    potential: np.ndarray
    width: int
    height: int
    fl: tuple
    pp: tuple
    true: list          # [(T_mc, reduced articulation)] per model
    init: list
    num_iterations: int


def barrett_reduction(hand):
    from .pose import LinearPoseReduction
    Fd, R = 8, 4
    A = np.zeros(Fd * R, dtype=F)
    for f, r in ((0, 0), (1, 1), (2, 1), (3, 0), (4, 2), (5, 2), (6, 3), (7, 3)):
        A[f * R + r] = 1
    lim = np.asarray(hand.joint_limits, dtype=F).reshape(-1, 2)
    # LinearPoseReduction::init copies the first redDims entries of the FULL joint limits (pose.cpp:58-64)
    return LinearPoseReduction(Fd, R, A, np.zeros(Fd, dtype=F), lim[:R, 0], lim[:R, 1], hand.joint_names[:R])


def config5_poses(frame, red):
    """True and start poses [(T_mc, reduced articulation)] x 2 models of frame `frame` of the config-5 sequence (seeded)."""
    rng = np.random.default_rng(SEED + 5 + frame)
    art = np.array([0.3, 0.9, 0.8, 1.0], dtype=F) * (0.8 + 0.4 * rng.random(4)).astype(F)
    art = np.clip(art, [red.getMin(i) for i in range(4)], [red.getMax(i) for i in range(4)]).astype(F)
    T_cm_h = se3.mul(se3.translation(-0.05, 0.02, 0.55), se3.exp(np.array([0, 0, 0, 1.3, 0.3, -0.2], dtype=F)))
    T_cm_o = se3.mul(se3.translation(0.12, -0.02, 0.60), se3.exp(np.array([0, 0, 0, 0.2, 0.5, 0.1], dtype=F)))
    true = [(se3.invert(T_cm_h), art), (se3.invert(T_cm_o), np.zeros(0, dtype=F))]
    init = []
    for T_mc, a in true:
        lim = np.stack([[red.getMin(i), red.getMax(i)] for i in range(len(a))]) if len(a) else None
        T, j = _perturb(T_mc, a.astype(np.float64), lim, rng, 0.012, 3.0, 0.08)
        init.append((T, j))
    return true, init


def config5(width=1280, height=720, frame=0):
    hand = load_asset("barrett").voxelize(0.005, 0.10)
    obj = _object_model()
    pot, _ = _hand_potential("barrett", len(hand.sdfs))
    red = barrett_reduction(hand)
    s = width / 640.0
    fl = (FOCAL * s, FOCAL * s)
    true, init = config5_poses(frame, red)
    return MultiScene("C5-barrett-object", hand, obj, red, pot, width, height, fl, (width / 2, height / 2), true, init, 5)


def config5_batch(tr, sc, frame_ids):
    """Synthetic observations and start poses of the frames `frame_ids` of the config-5 sequence for a tracker built from `sc`:
    (vert [F,H,W,4], norm [F,H,W,4] CUDA tensors, T_mcs [F,2,12], reduced articulations [F,4]). The observation of a frame is
    the ray-cast of its true pose, generated on the device."""
    import torch
    from .optimizer import synthetic_observation
    verts, norms, T0, a0 = [], [], [], []
    for k in frame_ids:
        true, init = config5_poses(int(k), sc.reduction)
        for m, (T, a) in enumerate(true):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
            tr.updatePose(m)
        _, v, n = synthetic_observation(tr.models, sc.width, sc.height, sc.fl, sc.pp)
        verts.append(v.clone())
        norms.append(n.clone())
        T0.append(np.stack([T for T, _ in init]))
        a0.append(np.concatenate([a for _, a in init]))
    return torch.stack(verts).contiguous(), torch.stack(norms).contiguous(), np.stack(T0).astype(F), np.stack(a0).astype(F)


def make_multi_tracker(sc):
    from .tracker import Tracker
    tr = Tracker(sc.width, sc.height, sc.fl[0])
    h = tr.addModel(sc.hand, reduction=sc.reduction, cloud_seed=21)
    tr.setIntersectionPotentialMatrix(h, sc.potential)
    ob = tr.addModel(sc.obj, cloud_seed=22)
    o = tr.getOptions()
    n = 2
    o.numIterations = sc.num_iterations
    o.focalLength = float(sc.fl[0])
    o.lambdaIntersection[h + n * h] = 5.0
    o.lambdaIntersection[ob + n * h] = 100.0
    o.lambdaIntersection[h + n * ob] = 100.0
    return tr
