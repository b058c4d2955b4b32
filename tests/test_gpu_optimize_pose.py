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
    from trajectory import segmented_parity
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
        return max(dt, dr, float(np.abs(a[1] - b[1]).max()) if len(a[1]) else 0.0)

    return segmented_parity(sc.name, iters, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3)


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


def test_optimize_pose_full_resolution_c2(gpu):
    """BASELINE config 2 at its full size (640x480, prediction 320x240): size-independent properties + oracle."""
    sc = configs.config2()
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = run_scene(sc, gpu, iters=5)
    dims = 22
    S = O.sys_len(dims)
    assert np.array_equal(g_info[:, 3], o_info[:, 3][:len(g_info)]) or abs(g_info[-1, 3] - o_info[-1, 3]) <= 0.002 * o_info[-1, 3]
    assert rel_err_blocks(g_first[:S], o_first[:S], dims) < 1e-4
    assert rel_err_blocks(g_first[S:2 * S], o_first[S:2 * S], dims) < 1e-4
    dt, dr = pose_diff(g_T, o_T)
    print("C2 full-size pose parity: |dt|=%.3e m |dr|=%.3e rad max|dq|=%.3e" % (dt, dr, np.abs(g_j - o_j).max()))
    assert dt < 1e-3 and dr < 1e-3 and np.abs(g_j - o_j).max() < 1e-3
    # energy decreases over the iterations
    assert g_info[-1, 0] < g_info[0, 0]
