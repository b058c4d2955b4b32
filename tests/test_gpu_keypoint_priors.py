"""GPU parity: Point3D3DPrior / Point2D3DPrior (point_3d_3d_prior.cpp, point_2d_3d_prior.cpp) inside the NoObs loop vs the
oracle's restatement, and the HandFitter multi-start fit (hand_fitter.cpp:273-356) as one batch. No reference test covers
these priors: parity unpinned (restatement only)."""
import numpy as np
import pytest

import oracle_lib as O
from dart_amd import configs, hand_fitter as HF, se3
from dart_amd.model import load_asset
from dart_amd.tracker_no_obs import ContactPrior, Point2D3DPrior, Point3D3DPrior, TrackerNoObs
from test_gpu_contact_grasp import sys_rel_err
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu


def hand_and_potential():
    hand = load_asset("human_hand").voxelize(0.005, 0.10)
    pot, _ = configs._hand_potential("human_hand", len(hand.sdfs))
    return hand, pot


def synthetic_keypoints(hand):
    """one outer point per finger link + three palm points (the reference's file needs mesh offsets that are not in its tree)"""
    rng = np.random.default_rng(5)
    kp = {}
    for name, frame in HF.GEOM_NAME2ID.items():
        if frame < hand.n_frames and frame != 0:
            kp[name] = (frame, [tuple(rng.uniform(-0.008, 0.008, 3)), (0.0, 0.0, 0.0)])
    for k in range(4):
        kp["palm%d" % k] = (0, [tuple(rng.uniform(-0.04, 0.04, 3)), (0.0, 0.0, 0.0)])
    return kp


def test_point_priors_system_and_trajectory_match_oracle(gpu):
    hand, pot = hand_and_potential()
    rng = np.random.default_rng(11)
    lim = np.asarray(hand.joint_limits, dtype=np.float64).reshape(-1, 2)
    tr = TrackerNoObs()
    tr.addModel(hand, cloud_seed=11)
    tr.setIntersectionPotentialMatrix(0, pot)
    o = tr.getOptions()
    o.lambdaIntersection = [20.0]
    o.regularization, o.regularizationScaled, o.numIterations = [0.1], [60.0], 4
    T_mc = se3.mul(se3.translation(0.02, -0.01, 0.5), se3.from_euler(0.2, -0.1, 0.3)).astype(F)
    joints = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(0.2, 0.6, len(lim))).astype(F)
    # targets: where the points of a different pose sit in the camera frame (3-D) / image (2-D), plus one zero-weight prior
    j_true = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(0.2, 0.6, len(lim))).astype(F)
    T_cm_true = se3.invert(se3.mul(se3.translation(0.03, 0.0, 0.52), se3.from_euler(0.25, -0.05, 0.2)).astype(F))
    T_mf_true = np.asarray(hand.forward_kinematics(j_true, "dh"), dtype=F).reshape(-1, 12)
    fl, pp = (537.0, 540.0), (320.0, 240.0)
    for k, frame in enumerate([0, 3, 4, 7, 8, 12, 16, 19, 20, 11]):
        pf = tuple(float(v) for v in rng.uniform(-0.01, 0.01, 3))
        pc = se3.apply(se3.mul(T_cm_true, T_mf_true[frame]), np.asarray([pf], dtype=F))[0]
        if k % 3 == 2:
            uv = (pc[0] * fl[0] / pc[2] + pp[0], pc[1] * fl[1] / pc[2] + pp[1])
            tr.addPrior(Point2D3DPrior(0, frame, uv, pf, 0.002, principalPoint=pp, focalLength=fl))
        else:
            tr.addPrior(Point3D3DPrior(0, frame, tuple(pc), pf, 20.0 if k else 0.0))
    assert len(tr.point_priors) == 10 and len(tr.priors) == 0
    D = hand.dims
    gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(T_mc[None, None], joints[None], debug=True)
    oms = [O.OModel(hand, 0)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    O.set_point_priors(tr.point_priors)
    try:
        oT, oj, oe, oinfo, ofirst = O.optimize_poses_noobs(oms, [T_mc], [joints], tr._options_struct(), clouds, pots, [])
        e0 = O.get_error_noobs(oms, [T_mc], [joints], tr._options_struct(), clouds, pots, [])
    finally:
        O.set_point_priors([])
    assert np.count_nonzero(ofirst) > 100
    err = sys_rel_err(gfirst[0], ofirst, D)
    dt, dr = pose_diff(gT[0, 0], oT[0])
    dj = float(np.abs(gj[0] - oj[0]).max())
    print("first system rel err %.2e; after 4 iterations |dt|=%.2e |dr|=%.2e max|dq|=%.2e; energy gpu %.6f oracle %.6f" % (err, dt, dr, dj, ge[0], oe))
    assert err < 1e-4
    assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3
    assert abs(ge[0] - oe) < 1e-3 * max(1.0, abs(oe))
    # the priors moved the pose, and getError (they carry no _error) equals the oracle's
    assert pose_diff(gT[0, 0], T_mc)[0] > 1e-3
    assert abs(tr.getError(T_mc[None, None], joints[None]) - e0) < 1e-3 * max(1.0, abs(e0))


def test_point_priors_next_to_contact_priors(gpu):
    """mixed list: the energy divides by the total number of priors (optimizer.cpp:1068) and both kinds enter the system"""
    sc = configs.contact_grasp_scene(2, n_attract=60, n_repulse=100)
    tr = configs.make_grasp_tracker(sc)
    tr.getOptions().numIterations = 2
    T, j = configs.grasp_batch_arrays(sc)
    T_mf = np.asarray(sc.hand.forward_kinematics(sc.gt_joints, "dh"), dtype=F).reshape(-1, 12)
    T_cm = se3.invert(sc.gt_T_mc)
    for frame in (4, 8, 20):
        pc = se3.apply(se3.mul(T_cm, T_mf[frame]), np.zeros((1, 3), dtype=F))[0]
        tr.addPrior(Point3D3DPrior(0, frame, tuple(pc), (0.0, 0.0, 0.0), 50.0))
    gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(T[1:2], j[1:2], debug=True)
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    O.set_point_priors(tr.point_priors)
    try:
        oT, oj, oe, oinfo, ofirst = O.optimize_poses_noobs(oms, [T[1, 0], T[1, 1]], [j[1], np.zeros(0, dtype=F)], tr._options_struct(), clouds,
                                                           pots, sc.priors)
    finally:
        O.set_point_priors([])
    assert sys_rel_err(gfirst[0], ofirst, 32) < 1e-4
    assert abs(ge[0] - oe) < 1e-3 * max(1.0, abs(oe))
    assert np.abs(gj[0] - oj[0]).max() < 1e-3
    # removing them again restores the contact-only energy normalisation
    tr.point_priors.clear()
    e_without = tr.getError(T[1:2], j[1:2])
    e_ref = O.get_error_noobs(oms, [T[1, 0], T[1, 1]], [j[1], np.zeros(0, dtype=F)], tr._options_struct(), clouds, pots, sc.priors)
    assert abs(e_without - e_ref) < 1e-3 * max(1.0, abs(e_ref))


def test_hand_fitter_multi_start_recovers_pose(gpu):
    hand, pot = hand_and_potential()
    kp = synthetic_keypoints(hand)
    rng = np.random.default_rng(2)
    lim = np.asarray(hand.joint_limits, dtype=np.float64).reshape(-1, 2)
    j_true = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(0.25, 0.6, len(lim))).astype(F)
    T_cm_true = se3.mul(se3.translation(0.05, -0.02, 0.6), se3.from_euler(0.3, 0.2, -0.4)).astype(F)
    T_mf = np.asarray(hand.forward_kinematics(j_true, "dh"), dtype=F).reshape(-1, 12)
    targets = {n: tuple(se3.apply(se3.mul(T_cm_true, T_mf[f]), np.asarray([pts[0]], dtype=F))[0]) for n, (f, pts) in kp.items()}
    inits = np.stack([(lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(0.0, 1.0, len(lim))).astype(F) for _ in range(15)]
                     + [np.zeros(len(lim), dtype=F)])
    fit = HF.HandFitter(hand, pot, kp, inits)
    fit.set_targets(targets, {n: False for n in targets})
    T = fit.init()
    # the rigid fit maps the camera-frame targets onto the zero-pose palm: a rotation (scale 1) for exact data
    assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-3
    T_mc, joints, err = fit.fit()
    print("multi-start errors (m): best %.5f, worst %.5f" % (min(fit.errors), max(fit.errors)))
    # GUI defaults (LM damping 60, regularisation 0.1, 40 iterations) converge slowly: the bar is a clear improvement over
    # every start, the best start being the one reported, and parity with the oracle below
    start = [fit.calculate_error(se3.identity().astype(F), inits[c]) for c in range(len(inits))]
    print("mean keypoint distance at the starts: best %.5f" % min(start))
    assert len(fit.errors) == 16 and err <= min(fit.errors) + 1e-9
    assert all(e < s0 for e, s0 in zip(fit.errors, start))
    # one start against the oracle
    tr = fit.tracker
    oms = [O.OModel(hand, 0)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    O.set_point_priors(tr.point_priors)
    try:
        oT, oj, oe, _, _ = O.optimize_poses_noobs(oms, [se3.identity().astype(F)], [inits[3]], tr._options_struct(), clouds, pots, [])
    finally:
        O.set_point_priors([])
    gT, gj, ge, _, _ = tr.optimizePosesBatch(se3.identity().astype(F)[None, None], inits[3][None])
    dt, dr = pose_diff(gT[0, 0], oT[0])
    print("start 3 after 40 iterations: |dt|=%.2e |dr|=%.2e max|dq|=%.2e" % (dt, dr, np.abs(gj[0] - oj[0]).max()))
    assert dt < 1e-3 and dr < 1e-3 and np.abs(gj[0] - oj[0]).max() < 2e-3
    # the NoObs solve is the reference's upper-triangular back substitution (SURVEY A-1), a damped fixed-point step rather
    # than a Newton step: convergence is slow by construction; more iterations keep closing the gap
    fit.set_max_iterations(200)
    _, _, err2 = fit.fit()
    print("200 iterations: best mean keypoint distance %.5f m (40 iterations: %.5f m)" % (err2, err))
    assert err2 < 0.8 * err
