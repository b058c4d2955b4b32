"""GPU parity: the device-resident Gauss-Newton loop (Optimizer::optimizePose) vs the CPU oracle's restatement
of optimizer.cpp:423-522. Bars: first-iteration systems within 1e-4 (norm-relative); poses within 1e-3 rad / 1e-3 m on every
segment of the trajectory (tests/trajectory.py: the oracle is restarted from the GPU pose at every discrete-decision flip --
no loosened bar anywhere), achieved values printed."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dart_amd import configs, se3
from dart_amd import device as D
from dart_amd.optimizer import Optimizer, synthetic_observation
from test_gpu_obs2mod import rel_err_blocks

F = np.float32
pytestmark = pytest.mark.gpu


def run_scene(sc, gpu, iters, lam_m2o=1.0, norm_thr=None):
    m = sc.host_model
    om = O.OModel(m, 0, sc.obs_layout)
    dm = D.DeviceModel(m, model_id=0, obs_layout=sc.obs_layout)
    dm.set_pose(sc.true_T_mc, sc.true_joints)
    depth, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
    sc.opts.numIterations = iters
    sc.opts.lambdaModToObs = lam_m2o
    if norm_thr is not None:
        sc.opts.normThreshold = norm_thr
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts, debug=True)
    g_T, _, _, g_j = dm.get_transforms()
    o_struct = sc.opts.c_struct(0, opt.pred_width, opt.pred_height, opt.pred_fl)
    o_T, o_j, o_info, o_first = O.optimize_pose(om, sc.init_T_mc, sc.init_joints, vert.cpu().numpy(), norm.cpu().numpy(), o_struct)
    return dm, g_T, g_j, opt.iter_info.cpu().numpy(), opt.first_system.cpu().numpy(), o_T, o_j, o_info, o_first


def pose_diff(Ta, Tb):
    """(translation distance [m], rotation angle [rad]) of Ta * Tb^-1, in float64."""
    D_ = se3.to44(Ta) @ np.linalg.inv(se3.to44(Tb))
    c = np.clip((np.trace(D_[:3, :3]) - 1) / 2, -1, 1)
    ang = np.arccos(c) if c < 1 - 1e-12 else np.linalg.norm(D_[:3, :3] - np.eye(3)) / np.sqrt(2)
    return float(np.linalg.norm(D_[:3, 3])), float(ang)


def segmented_pose_parity(sc0, cfg, size, gpu, iters, lam_m2o=1.0):
    """tests/trajectory.py for Optimizer::optimizePose: GPU prefixes from the common start vs the oracle restarted from the
    GPU's pose at every decision flip (observation SDF kept at the run's start pose, as the original call has it)."""
    from trajectory import parts, segmented_parity
    sc = cfg(*size)
    m = sc.host_model
    om = O.OModel(m, 0, sc.obs_layout)
    dm = D.DeviceModel(m, model_id=0, obs_layout=sc.obs_layout)
    dm.set_pose(sc.true_T_mc, sc.true_joints)
    _, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
    v_np, n_np = vert.cpu().numpy(), norm.cpu().numpy()
    sc.opts.lambdaModToObs = lam_m2o
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)

    def gpu_prefix(k):
        sc.opts.numIterations = k
        dm.set_pose(sc.init_T_mc, sc.init_joints)
        opt.optimizePose(dm, vert, norm, sc.opts, debug=True)
        T, _, _, j = dm.get_transforms()
        return (T.copy(), j.copy()), opt.iter_info.cpu().numpy().copy()

    def oracle_from(pose, k0, n):
        sc.opts.numIterations = n
        o_struct = sc.opts.c_struct(0, opt.pred_width, opt.pred_height, opt.pred_fl)
        with O.obs_sdf_start(sc.init_T_mc):
            T, j, info, _ = O.optimize_pose(om, pose[0], pose[1], v_np, n_np, o_struct)
        return (T, j), info

    def decisions_equal(g, o):
        return bool(np.all(g[3:5] == o[3:5]) and np.all(np.abs(g[0:2] - o[0:2]) <= 2e-5 * np.abs(o[0:2]) + 1e-9))

    def pose_err(a, b):
        dt, dr = pose_diff(a[0], b[0])
        return parts(translation_m=[dt], rotation_rad=[dr], joints_rad=[np.abs(a[1] - b[1]).max()] if len(a[1]) else [])

    def near_ties(gpose, opose, k):
        import near_ties as NT
        recs = NT.association_near_ties(om, v_np, n_np, sc.opts, gpose, opose)
        NT.assert_explained(recs, "%s iteration %d" % (sc.name, k))
        return recs

    return segmented_parity("%s %dx%d" % (sc.name, sc.width, sc.height), iters, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3,
                            near_ties=near_ties)


@pytest.mark.parametrize("cfg,size", [(configs.config1, (320, 240)), (configs.config2, (320, 240)), (configs.config2_left_hand, (320, 240))])
def test_optimize_pose_matches_oracle(gpu, cfg, size):
    sc = cfg(*size)
    dims = 6 + sc.host_model.n_joints
    S = O.sys_len(dims)
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = run_scene(sc, gpu, iters=5)
    # first iteration: identical association counts and systems within tolerance
    assert g_info[0, 3] == o_info[0, 3] and g_info[0, 3] > 1000
    assert g_info[0, 4] == o_info[0, 4] and g_info[0, 4] > 200
    assert rel_err_blocks(g_first[:S], o_first[:S], dims) < 1e-4
    assert rel_err_blocks(g_first[S:2 * S], o_first[S:2 * S], dims) < 1e-4
    # converged pose: every segment of the trajectory between discrete-decision flips is held to 1e-3 (tests/trajectory.py)
    dt, dr = pose_diff(g_T, o_T)
    dj = float(np.abs(g_j - o_j).max()) if len(g_j) else 0.0
    print("free-running pose parity %s: |dt|=%.3e m |dr|=%.3e rad max|dq|=%.3e rad" % (sc.name, dt, dr, dj))
    segs = segmented_pose_parity(sc, cfg, size, gpu, 5)
    if len(segs) == 1:
        assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3
    # and the optimisation actually did its job: the translation error and the total energy went down
    # (rotation is unobservable for the sphere and weakly observable for proxy hands in 5 damped iterations)
    t0, _ = pose_diff(sc.init_T_mc, sc.true_T_mc)
    t1, _ = pose_diff(g_T, sc.true_T_mc)
    assert t1 < t0
    assert g_info[-1, 0] + g_info[-1, 1] < g_info[0, 0] + g_info[0, 1]


def test_optimize_pose_reference_default_normal_gate(gpu):
    """normThreshold = -0.1 (optimization.h:44): the inverted gate keeps only a few percent of the points; parity holds."""
    sc = configs.config2(320, 240)
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = run_scene(sc, gpu, iters=3, norm_thr=-0.1)
    S = O.sys_len(22)
    assert g_info[0, 3] == o_info[0, 3] and 50 < g_info[0, 3] < 2000
    assert rel_err_blocks(g_first[:S], o_first[:S], 22) < 1e-4
    dt, dr = pose_diff(g_T, o_T)
    assert dt < 1e-3 and dr < 1e-3 and np.abs(g_j - o_j).max() < 1e-3


def test_optimize_pose_obs_to_mod_only_and_zero_iterations(gpu):
    sc = configs.config2(320, 240)
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = run_scene(sc, gpu, iters=3, lam_m2o=0.0)
    dt, dr = pose_diff(g_T, o_T)
    assert dt < 1e-3 and dr < 1e-3 and np.abs(g_j - o_j).max() < 1e-3
    assert g_info[0, 4] == 0
    dm2, g_T, g_j, *_ = run_scene(configs.config2(160, 120), gpu, iters=0)
    assert np.array_equal(g_T, configs.config2(160, 120).init_T_mc)


def test_association_flips_under_a_small_pose_change_are_near_ties(gpu):
    """The documentation north_star asks for, on a case that HAS flips: the same 640x480 frame associated at a pose and at that
    pose moved by 2e-5 m and 1e-5 rad (two summation orders differ by 1e-7 and flip nothing here; the larger step makes the list
    non-empty). GPU and oracle agree bit for bit at each pose, so the
    pixels that flip are the same on both; each is listed with its margin and must have its threshold between the two poses'
    values. The list goes to the parity records (tests/golden/parity_records/)."""
    import near_ties as NT
    from trajectory import store_record
    sc = configs.config2()
    m = sc.host_model
    om = O.OModel(m, 0, sc.obs_layout)
    dm = D.DeviceModel(m, model_id=0, obs_layout=sc.obs_layout)
    dm.set_pose(sc.true_T_mc, sc.true_joints)
    _, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
    v_np, n_np = vert.cpu().numpy(), norm.cpu().numpy()
    sc.opts.normThreshold = -0.1            # the reference's default gate: both criteria can flip
    sc.opts.distThreshold = [0.004]         # a threshold that cuts through the start pose's residuals (1.5 cm off): many pixels near it
    pose_a = (sc.init_T_mc.copy(), sc.init_joints.copy())
    pose_b = (se3.mul(se3.exp(np.array([2e-5, -1e-5, 1.5e-5, 1e-5, 0, -1e-5], dtype=F)), sc.init_T_mc), sc.init_joints.copy())
    das = []
    for T, j in (pose_a, pose_b):
        dm.set_pose(T, j)
        pts, count, dbg = D.error_and_data_association(vert, norm, dm, dist_threshold=sc.opts.distThreshold[0], norm_threshold=-0.1, debug=True)
        das.append(dbg["da"].cpu().numpy().reshape(-1).copy())
    recs = NT.association_near_ties(om, v_np, n_np, sc.opts, pose_a, pose_b)
    gpu_flips = set(np.nonzero(das[0] != das[1])[0].tolist())
    assert gpu_flips == {r["pixel"] for r in recs}, "GPU and oracle disagree on which pixels flip"
    NT.assert_explained(recs, "C2 640x480 association under a 2e-5 m pose change")
    n_assoc = int((das[0] >= 0).sum())
    print("association of %d pixels at two poses 2e-5 m apart: %d flip, margins %s" % (n_assoc, len(recs), sorted(r["margin"] for r in recs)[:6]))
    assert n_assoc > 1000 and 0 < len(recs) < n_assoc // 20
    store_record("C2 640x480 association near-ties under a 2e-5 m pose change",
                 {"associated": n_assoc, "flipped": len(recs), "near_ties": recs,
                  "largest_margin": max([r["margin"] for r in recs], default=0.0)})


def test_optimize_pose_full_resolution_c2(gpu):
    """BASELINE config 2 at its full size (640x480, prediction 320x240): size-independent properties + oracle."""
    sc = configs.config2()
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = run_scene(sc, gpu, iters=5)
    dims = 22
    S = O.sys_len(dims)
    # counts: equal in every iteration, or the run's flips are listed and shown to be near-ties by the segmented comparison below
    counts_equal = bool(np.array_equal(g_info[:, 3:5], o_info[:len(g_info), 3:5]))
    segs = segmented_pose_parity(sc, configs.config2, (640, 480), gpu, 5)
    assert counts_equal == (len(segs) == 1)
    assert rel_err_blocks(g_first[:S], o_first[:S], dims) < 1e-4
    assert rel_err_blocks(g_first[S:2 * S], o_first[S:2 * S], dims) < 1e-4
    dt, dr = pose_diff(g_T, o_T)
    print("C2 full-size pose parity: |dt|=%.3e m |dr|=%.3e rad max|dq|=%.3e" % (dt, dr, np.abs(g_j - o_j).max()))
    assert dt < 1e-3 and dr < 1e-3 and np.abs(g_j - o_j).max() < 1e-3
    # energy decreases over the iterations
    assert g_info[-1, 0] < g_info[0, 0]
