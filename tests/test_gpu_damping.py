"""GPU parity: the trackers' damping matrices (Tracker::getDampingMatrix, tracker.h:122; TrackerNoObs::getDampingMatrix,
tracker_no_obs.h:73). The multi-model loop adds a model's matrix to its block after the regularisation and the LM scaling
(optimizer.cpp:702-703), the NoObs loop before the priors contribute (optimizer.cpp:895-896). Both trackers start them as zero
(tracker.cpp:246-247, tracker_no_obs.cpp:214-215) and no test of the reference sets one: parity is pinned by the oracle's
restatement only ("parity unpinned")."""
import numpy as np
import pytest

import oracle_lib as O
from dart_amd import configs, se3
from test_gpu_optimize_pose import pose_diff
from test_gpu_contact_grasp import scene, sys_rel_err
from test_gpu_multi_model import setup

F = np.float32
pytestmark = pytest.mark.gpu


def spd(dims, scale, seed):
    """a symmetric positive definite matrix with off-diagonal entries (so that the upper-triangle handling is exercised)"""
    r = np.random.RandomState(seed)
    B = r.standard_normal((dims, dims))
    return (scale * (B @ B.T / dims + np.eye(dims))).astype(F)


def test_noobs_damping_matches_oracle_and_changes_the_step(gpu):
    sc = scene(4)
    T, j = configs.grasp_batch_arrays(sc)
    c = 1
    results = {}
    for damped in (False, True):
        tr = configs.make_grasp_tracker(sc)
        tr.getOptions().numIterations = 3
        mats = [None, None]
        if damped:
            Dh, Do = spd(32 - 6, 5.0, 1), spd(6, 2.0, 2)          # hand block 26 x 26, object block 6 x 6
            tr.getDampingMatrix(0)[:] = Dh
            tr.getDampingMatrix(1)[:] = Do
            mats = [Dh, Do]
        assert tr.getDampingMatrix(0).shape == (26, 26) and tr.getDampingMatrix(1).shape == (6, 6)
        gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(T[c:c + 1], j[c:c + 1], debug=True)
        oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
        o = tr._options_struct()
        clouds = [x.cpu().numpy() for x in tr.clouds]
        pots = [x.cpu().numpy() for x in tr.potentials]
        with O.damping_matrices(mats):
            oT, oj, oe, oinfo, ofirst = O.optimize_poses_noobs(oms, [T[c, 0], se3.identity()], [j[c], np.zeros(0, dtype=F)], o, clouds, pots,
                                                               sc.priors)
        assert sys_rel_err(gfirst[0], ofirst, 32) < 1e-4
        dt, dr = pose_diff(gT[0, 0], oT[0])
        dto, dro = pose_diff(gT[0, 1], oT[1])
        dj = float(np.abs(gj[0] - oj[0]).max())
        print("NoObs damping %s: hand |dt|=%.2e |dr|=%.2e max|dq|=%.2e object |dt|=%.2e |dr|=%.2e energy gpu %.6f oracle %.6f"
              % ("on" if damped else "off", dt, dr, dj, dto, dro, ge[0], oe))
        assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3 and dto < 1e-3 and dro < 1e-3
        assert abs(ge[0] - oe) < 1e-3 * max(1.0, abs(oe))
        results[damped] = (gT.copy(), gj.copy(), gfirst[0].copy())
        tr.close()
    # the first system carries the matrices (it is taken before the regularisation), and the damped step is shorter
    d_first = (results[True][2][:32 * 32] - results[False][2][:32 * 32]).reshape(32, 32)
    assert np.allclose(d_first[:26, :26], np.triu(spd(26, 5.0, 1)), rtol=1e-3, atol=1e-3)
    assert np.allclose(d_first[26:, 26:], np.triu(spd(6, 2.0, 2)), rtol=1e-3, atol=1e-3)
    move_plain = float(np.abs(results[False][1][0] - j[c]).max())
    move_damped = float(np.abs(results[True][1][0] - j[c]).max())
    print("largest joint move after 3 iterations: %.4f without, %.4f with damping" % (move_plain, move_damped))
    assert move_damped < move_plain


def test_noobs_damping_can_be_switched_off_again(gpu):
    sc = scene(4)
    T, j = configs.grasp_batch_arrays(sc)
    tr = configs.make_grasp_tracker(sc)
    tr.getOptions().numIterations = 2
    ref = tr.optimizePosesBatch(T[:2], j[:2])
    tr.getDampingMatrix(0)[:] = spd(26, 5.0, 3)
    damped = tr.optimizePosesBatch(T[:2], j[:2])
    assert float(np.abs(damped[1] - ref[1]).max()) > 1e-4
    tr.getDampingMatrix(0)[:] = 0
    again = tr.optimizePosesBatch(T[:2], j[:2])
    assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
    tr.close()


@pytest.mark.parametrize("device_loop", [False, True])
def test_multi_model_damping_matches_oracle(gpu, device_loop):
    sc, tr, vert, norm = setup(320, 180, 3)
    tr.getDampingMatrix(0)[:] = spd(10, 50.0, 4)          # Barrett, 6 + 4 reduced dimensions
    tr.getDampingMatrix(1)[:] = spd(6, 20.0, 5)
    (tr.optimizePosesDevice if device_loop else tr.optimizePoses)(vert, norm, debug=True)
    gfirst = tr.first_system.copy()
    oT, oa, oinfo, ofirst = O.optimize_poses(tr, vert.cpu().numpy(), norm.cpu().numpy(), [T for T, _ in sc.init], [a for _, a in sc.init])
    assert np.array_equal(tr.iteration_summaries[0, :, 3:5], oinfo[0, :, 3:5])
    assert sys_rel_err(gfirst, ofirst, 16) < 1e-4
    for m in range(2):
        dt, dr = pose_diff(tr.getPose(m).getTransformCameraToModel(), oT[m])
        da = float(np.abs(tr.getPose(m).getReducedArticulation() - oa[m]).max()) if len(oa[m]) else 0.0
        print("C5 damping, %s loop, model %d: |dt|=%.2e |dr|=%.2e max|da|=%.2e" % ("device" if device_loop else "host-driven", m, dt, dr, da))
        assert dt < 1e-3 and dr < 1e-3 and da < 1e-3
    # and the matrices are in the system: the same run without them differs in the first system by exactly their upper triangles
    sc2, tr2, vert2, norm2 = setup(320, 180, 1)
    (tr2.optimizePosesDevice if device_loop else tr2.optimizePoses)(vert2, norm2, debug=True)
    d = (gfirst[:256] - tr2.first_system[:256]).reshape(16, 16)
    assert np.allclose(d[:10, :10], np.triu(spd(10, 50.0, 4)), rtol=1e-3, atol=2e-2)
    assert np.allclose(d[10:, 10:], np.triu(spd(6, 20.0, 5)), rtol=1e-3, atol=2e-2)


def test_batch_of_frames_takes_the_damping_matrices(gpu):
    """dartk_optimize_poses_batch shares the models' options, the damping matrices among them: per frame the result equals the
    single-frame device loop's."""
    from test_gpu_frame_batch import frames, single
    tr, scs, vert, norm, T0, a0 = frames(320, 180, (0, 3), 3)
    plain_T, _ = tr.optimizePosesBatch(vert, norm, T0, a0)
    tr.getDampingMatrix(0)[:] = spd(10, 50.0, 4)
    bT, ba = tr.optimizePosesBatch(vert, norm, T0, a0)
    assert float(np.abs(bT - plain_T).max()) > 1e-5
    for i in range(2):
        sT, sa, _ = single(tr, scs[i], vert[i], norm[i])
        for m in range(2):
            dt, dr = pose_diff(sT[m], bT[i, m])
            assert dt < 1e-4 and dr < 1e-3, (i, m, dt, dr)       # the bars of test_batch_equals_single_frame_loop
        assert float(np.abs(sa - ba[i][:4]).max()) < 1e-3
