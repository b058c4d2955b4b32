"""GPU parity: multi-model tracking with observations and a reduced pose (BASELINE config 5: Barrett 8->4 + object;
Optimizer::optimizePoses, optimizer.cpp:524-781) vs the CPU oracle. Every term goes through the C ABI; the loop's host
part (assembly, triangular solve, update) is dart_amd/tracker.py."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dart_amd import configs
from dart_amd.optimizer import synthetic_observation
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu


def setup(width, height, iters, frame=0):
    sc = configs.config5(width, height, frame)
    tr = configs.make_multi_tracker(sc)
    for m, (T, a) in enumerate(sc.true):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
        tr.updatePose(m)
    _, vert, norm = synthetic_observation(tr.models, sc.width, sc.height, sc.fl, sc.pp)
    tr.getOptions().numIterations = iters
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    return sc, tr, vert, norm


def run(width, height, iters, frame=0, device_loop=False):
    sc, tr, vert, norm = setup(width, height, iters, frame)
    (tr.optimizePosesDevice if device_loop else tr.optimizePoses)(vert, norm, debug=True)
    oT, oa, oinfo, ofirst = O.optimize_poses(tr, vert.cpu().numpy(), norm.cpu().numpy(), [T for T, _ in sc.init], [a for _, a in sc.init])
    return sc, tr, vert, norm, oT, oa, oinfo, ofirst


def segmented(sc, tr, vert, norm, iters, device_loop):
    """tests/trajectory.py for Optimizer::optimizePoses: every segment between count flips held to 1e-3, the oracle restarted
    from the GPU's poses (observation SDFs kept at the run's start poses)."""
    from trajectory import parts, segmented_parity
    v_np, n_np = vert.cpu().numpy(), norm.cpu().numpy()
    init_T = [T for T, _ in sc.init]

    def gpu_prefix(k):
        tr.getOptions().numIterations = k
        for m, (T, a) in enumerate(sc.init):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
        (tr.optimizePosesDevice if device_loop else tr.optimizePoses)(vert, norm, debug=True)
        pose = [(tr.getPose(m).getTransformCameraToModel().copy(), tr.getPose(m).getReducedArticulation().copy()) for m in range(2)]
        return pose, tr.iteration_summaries.copy()

    def oracle_from(pose, k0, n):
        tr.getOptions().numIterations = n
        with O.obs_sdf_start(init_T):
            oT, oa, oinfo, _ = O.optimize_poses(tr, v_np, n_np, [p[0] for p in pose], [p[1] for p in pose])
        return [(oT[m], oa[m]) for m in range(2)], oinfo

    def decisions_equal(g, o):
        return bool(np.array_equal(g[:, 3:5], o[:, 3:5]))

    def pose_err(a, b):
        d = [pose_diff(a[m][0], b[m][0]) for m in range(2)]
        return parts(translation_m=[x[0] for x in d], rotation_rad=[x[1] for x in d],
                     joints_rad=[np.abs(a[m][1] - b[m][1]).max() for m in range(2) if len(b[m][1])])

    name = "C5 %dx%d %s loop" % (sc.width, sc.height, "device" if device_loop else "host-driven")
    segs = segmented_parity(name, iters, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3)
    tr.getOptions().numIterations = iters
    return segs


def check(sc, tr, vert, norm, oT, oa, oinfo, ofirst, device_loop=False):
    ginfo = tr.iteration_summaries
    iters = len(ginfo)
    # iteration 0: same association / prediction counts per model, same system
    assert np.array_equal(ginfo[0, :, 3:5], oinfo[0, :, 3:5]), (ginfo[0], oinfo[0])
    assert ginfo[0, 0, 3] > 200 and ginfo[0, 1, 3] > 100 and ginfo[0, 0, 4] > 50
    D = 16
    for sl in (slice(0, D * D), slice(D * D, D * D + D)):
        nref = np.linalg.norm(ofirst[sl].astype(np.float64))
        assert np.linalg.norm(tr.first_system[sl].astype(np.float64) - ofirst[sl]) / nref < 1e-4
    for m in range(2):
        moved = float(np.abs(oT[m] - sc.init[m][0]).max())
        assert moved > 1e-4
    # every segment of the trajectory between count flips within 1e-3 (no loosened bar)
    segmented(sc, tr, vert, norm, iters, device_loop)


def test_config5_small_matches_oracle(gpu):
    check(*run(320, 180, 4))


def test_config5_moves_towards_truth_and_respects_reduced_limits(gpu):
    sc, tr, vert, norm = setup(320, 180, 5)
    t0 = [pose_diff(sc.init[m][0], sc.true[m][0])[0] for m in range(2)]
    tr.optimizePoses(vert, norm)
    t1 = [pose_diff(tr.getPose(m).getTransformCameraToModel(), sc.true[m][0])[0] for m in range(2)]
    print("translation error before %s after %s" % (t0, t1))
    assert t1[0] < t0[0] and t1[1] < t0[1]
    ra = tr.getPose(0).getReducedArticulation()
    for i in range(4):
        assert tr.getPose(0).getReducedMin(i) <= ra[i] <= tr.getPose(0).getReducedMax(i)
    # coupled joints stay coupled (A maps reduced dof 1 to full joints 1 and 2, grasp_analyzer.cpp:411-412)
    full = tr.getPose(0).getArticulation()
    assert full[1] == full[2] and full[0] == full[3] and full[4] == full[5] and full[6] == full[7]


def test_config5_full_resolution(gpu):
    """BASELINE config 5 at its full 1280x720: counts and first system against the oracle, all 5 iterations of the
    configuration segment by segment, host-driven and device-resident loops."""
    check(*run(1280, 720, 5))
    check(*run(1280, 720, 5, device_loop=True), device_loop=True)


def test_device_loop_matches_host_driven_loop_and_oracle(gpu):
    """dartk_optimize_poses (whole loop on the device) vs the host-driven loop over the same term kernels and vs the oracle.
    The association lists are appended with atomics (as in the reference), so sums differ in order: agreement to rounding."""
    iters = 4
    sc, tr, vert, norm = setup(320, 180, iters)
    tr.optimizePoses(vert, norm, debug=True)
    host_T = [tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)]
    host_a = [tr.getPose(m).getReducedArticulation().copy() for m in range(2)]
    host_info, host_first = tr.iteration_summaries.copy(), tr.first_system.copy()
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, debug=True)
    D = 16
    for sl in (slice(0, D * D), slice(D * D, D * D + D)):
        nref = np.linalg.norm(host_first[sl].astype(np.float64))
        assert np.linalg.norm(tr.first_system[sl].astype(np.float64) - host_first[sl]) / nref < 1e-5
    assert np.array_equal(tr.iteration_summaries[0, :, 3:5], host_info[0, :, 3:5])
    assert np.allclose(tr.iteration_summaries[0, :, :3], host_info[0, :, :3], rtol=1e-4, atol=1e-7)
    same_counts = np.array_equal(tr.iteration_summaries[:, :, 3:5], host_info[:, :, 3:5])
    dev_T = [tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)]
    dev_a = [tr.getPose(m).getReducedArticulation().copy() for m in range(2)]
    for m in range(2):
        dt, dr = pose_diff(dev_T[m], host_T[m])
        da = float(np.abs(dev_a[m] - host_a[m]).max()) if len(host_a[m]) else 0.0
        print("model %d device loop vs host loop: |dt|=%.2e |dr|=%.2e max|dq_red|=%.2e (counts equal: %s)" % (m, dt, dr, da, same_counts))
        if same_counts:     # same discrete decisions: the two loops are the same arithmetic up to the order of the appended lists
            assert dt < 1e-4 and dr < 1e-3 and da < 1e-3
    # and the oracle, segment by segment
    oT, oa, oinfo, ofirst = O.optimize_poses(tr, vert.cpu().numpy(), norm.cpu().numpy(), [T for T, _ in sc.init], [a for _, a in sc.init])
    check(sc, tr, vert, norm, oT, oa, oinfo, ofirst, device_loop=True)
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, debug=True)
    # the full joint vector on the device is the projection of the reduced one (Barrett: 8 joints from 4)
    _, _, _, joints = tr.models[0].get_transforms()
    assert np.allclose(joints, tr.getPose(0).getArticulation(), atol=1e-6)


def test_device_loop_repeats(gpu):
    """The three chains of an iteration (ray-cast + mod->obs, association + obs->mod, intersection terms) run on three
    streams and share the slab buffer by lanes: repeated runs from the same start agree to the rounding of the atomically
    appended association lists (a race between the chains would not)."""
    sc, tr, vert, norm = setup(640, 360, 6)
    runs = []
    for _ in range(5):
        for m, (T, a) in enumerate(sc.init):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
        tr.optimizePosesDevice(vert, norm, debug=True)
        runs.append(([tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)],
                     tr.getPose(0).getReducedArticulation().copy(), tr.iteration_summaries.copy()))
    for T, a, info in runs[1:]:
        # iteration 0 starts from identical inputs: exact counts. Later iterations start from poses that agree only to the
        # rounding of the atomically appended lists, so a pixel sitting on a threshold may flip: a few points of tolerance.
        assert np.array_equal(info[0, :, 3:5], runs[0][2][0, :, 3:5])
        assert np.abs(info[:, :, 3:5] - runs[0][2][:, :, 3:5]).max() <= 3
        for m in range(2):
            assert np.abs(T[m] - runs[0][0][m]).max() < 2e-5      # measured 2e-6 on one stream and on three alike
        assert np.abs(a - runs[0][1]).max() < 1e-4                  # measured 1.6e-5


def test_shared_launch_form_equals_three_stream_form(gpu, monkeypatch):
    """The single-frame device loop runs its terms in shared launches on one stream (union_terms_kernel) when the problem is
    eligible, else on three streams with one launch per term: the same bodies on the same data, only the number of workgroups
    per term -- the partition of the fixed-order sums -- differs. Counts must be equal in every iteration, energies and poses
    agree to rounding; and a call with readback=False leaves the same result on the device as the default call returns."""
    iters = 6
    sc, tr, vert, norm = setup(640, 360, iters)

    def run_once(**kw):
        for m, (T, a) in enumerate(sc.init):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
        tr.optimizePosesDevice(vert, norm, debug=True, **kw)
        pose = [(tr.getPose(m).getTransformCameraToModel().copy(), tr.getPose(m).getReducedArticulation().copy()) for m in range(2)]
        return pose, tr.iteration_summaries.copy()

    pu, iu = run_once()
    monkeypatch.setenv("DARTK_MULTI_NO_UNION", "1")
    ps, is_ = run_once()
    monkeypatch.delenv("DARTK_MULTI_NO_UNION")
    assert np.array_equal(iu[:, :, 3:5], is_[:, :, 3:5]) and iu[:, :, 3].min() > 100
    assert np.all(np.abs(iu[:, :, :3] - is_[:, :, :3]) <= 2e-4 * np.abs(is_[:, :, :3]) + 1e-7)       # (later iterations start from poses that differ by rounding)
    assert np.all(np.abs(iu[0, :, :3] - is_[0, :, :3]) <= 2e-6 * np.abs(is_[0, :, :3]) + 1e-9)
    for m in range(2):
        dt, dr = pose_diff(pu[m][0], ps[m][0])
        # (the box's rotation is weakly observed: rounding-sized differences of the sums grow to a few 1e-4 rad in six iterations)
        assert dt < 1e-5 and dr < 1e-3 and (len(pu[m][1]) == 0 or np.abs(pu[m][1] - ps[m][1]).max() < 1e-3), (m, dt, dr)
    # poses left on the device by a call that does not read back
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, readback=False)
    torch.cuda.synchronize()
    for m in range(2):      # (to rounding: the association lists are appended in a different order run to run)
        dt, dr = pose_diff(tr.models[m].get_T_mc(), pu[m][0])
        assert dt < 1e-5 and dr < 1e-3, (m, dt, dr)
        n_red = len(pu[m][1])
        if n_red:
            assert np.abs(tr._red_art_dev[m].cpu().numpy()[:n_red] - pu[m][1]).max() < 1e-3


def test_prediction_map_of_the_fused_replay_is_the_replay_launch_s(gpu, monkeypatch):
    """In the shared-launch form the mod->obs terms form the prediction themselves (replay of every model's traces in model
    order, then cullUnobservable; the first model's term writes the map). After ONE iteration from the same start poses the
    map the caller gets must be, bit for bit, the map of the replay + cull launch -- in the shared-launch form with that launch
    kept (DARTK_MULTI_REPLAY_LAUNCH) and in the three-stream form (DARTK_MULTI_NO_UNION) -- and so must the mod->obs counts."""
    sc, tr, vert, norm = setup(640, 360, 1)

    def run_once():
        for m, (T, a) in enumerate(sc.init):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
        tr.prediction.zero_()
        tr.optimizePosesDevice(vert, norm, debug=True)
        return tr.prediction.cpu().numpy().copy(), tr.iteration_summaries.copy()

    fused, i_f = run_once()
    monkeypatch.setenv("DARTK_MULTI_REPLAY_LAUNCH", "1")
    launch, i_l = run_once()
    monkeypatch.delenv("DARTK_MULTI_REPLAY_LAUNCH")
    monkeypatch.setenv("DARTK_MULTI_NO_UNION", "1")
    streams, i_s = run_once()
    monkeypatch.delenv("DARTK_MULTI_NO_UNION")
    hits = fused[..., 2] != 0
    assert hits.sum() > 500 and len(np.unique(fused[..., 3][hits].astype(np.int64) >> 16)) == 2       # both models are seen
    assert fused.tobytes() == launch.tobytes()
    assert fused.tobytes() == streams.tobytes()
    assert np.array_equal(i_f[:, :, 4], i_l[:, :, 4]) and np.array_equal(i_f[:, :, 4], i_s[:, :, 4]) and i_f[0, :, 4].min() > 50


def test_tracking_from_the_poses_left_on_the_device(gpu):
    """Three frames in a row, each starting where the one before ended: with every call reading the poses back and uploading
    them again, and with the poses never leaving the device (readback=False, then from_device=True; readPoses() at the end).
    The uploads carry the same bits the device held, so the two must agree to the rounding the association lists' run-to-run
    order allows."""
    sc, tr, vert, norm = setup(640, 360, 4)

    def start():
        for m, (T, a) in enumerate(sc.init):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a

    def poses():
        return [(tr.getPose(m).getTransformCameraToModel().copy(), tr.getPose(m).getReducedArticulation().copy()) for m in range(2)]

    start()
    for _ in range(3):
        tr.optimizePosesDevice(vert, norm)
    host = poses()
    start()
    tr.optimizePosesDevice(vert, norm, readback=False)
    for _ in range(2):
        tr.optimizePosesDevice(vert, norm, readback=False, from_device=True)
    stale = poses()
    for m, (T, a) in enumerate(sc.init):        # nothing has come back yet: the host still holds the start poses
        assert np.array_equal(stale[m][0], np.asarray(T, dtype=np.float32).reshape(stale[m][0].shape))
    tr.readPoses()
    dev = poses()
    moved = 0.0
    for m in range(2):
        dt, dr = pose_diff(dev[m][0], host[m][0])
        assert dt < 1e-5 and dr < 1e-3, (m, dt, dr)
        if len(host[m][1]):
            assert np.abs(dev[m][1] - host[m][1]).max() < 1e-3
        moved = max(moved, pose_diff(dev[m][0], stale[m][0])[0])
    assert moved > 1e-4                         # the frames did move the poses
    tr2 = setup(640, 360, 4)[1]
    with pytest.raises(ValueError):
        tr2.optimizePosesDevice(vert, norm, from_device=True)

