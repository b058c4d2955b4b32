"""GPU parity of the batched-frames loop (dartk_optimize_poses_batch, BASELINE configs[4]: a batch of independent Barrett + object
frames): every frame of a batch equals the single-frame device loop on that frame (and through it the oracle, which
test_gpu_multi_model.py pins), distinct frames give distinct results, a frame's result does not depend on the batch it is in,
and one frame is held against the oracle directly."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dart_amd import configs
from dart_amd.optimizer import synthetic_observation
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu


def frames(width, height, ks, iters):
    """Tracker + per-frame observations / start poses of config5(frame=k) for k in ks (same models and options for all k)."""
    tr = None
    verts, norms, T0, a0, scs = [], [], [], [], []
    for k in ks:
        sc = configs.config5(width, height, k)
        if tr is None:
            tr = configs.make_multi_tracker(sc)
        for m, (T, a) in enumerate(sc.true):
            tr.getPose(m).setTransformCameraToModel(T)
            tr.getPose(m).getReducedArticulation()[:] = a
            tr.updatePose(m)
        _, v, n = synthetic_observation(tr.models, sc.width, sc.height, sc.fl, sc.pp)
        verts.append(v.clone()); norms.append(n.clone())
        T0.append(np.stack([T for T, _ in sc.init]))
        a0.append(np.concatenate([a for _, a in sc.init]))
        scs.append(sc)
    tr.getOptions().numIterations = iters
    return tr, scs, torch.stack(verts).contiguous(), torch.stack(norms).contiguous(), np.stack(T0).astype(F), np.stack(a0).astype(F)


def single(tr, sc, vert, norm):
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, debug=True)
    return ([tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)], tr.getPose(0).getReducedArticulation().copy(),
            tr.iteration_summaries.copy())


@pytest.mark.parametrize("size,ks,iters", [((320, 180), (0, 1, 2, 3, 4, 5, 6), 4), ((1280, 720), (0, 3, 9), 3)])
def test_batch_equals_single_frame_loop(gpu, size, ks, iters):
    tr, scs, vert, norm, T0, a0 = frames(size[0], size[1], ks, iters)
    bT, ba = tr.optimizePosesBatch(vert, norm, T0, a0, debug=True)
    binfo = tr.iteration_summaries.copy()
    assert np.isfinite(bT).all() and np.isfinite(ba).all()
    worst = 0.0
    for i, k in enumerate(ks):
        sT, sa, sinfo = single(tr, scs[i], vert[i], norm[i])
        # iteration 0 starts from identical inputs: identical counts; later iterations start from poses equal to the rounding of
        # the atomically appended association lists (as between two single calls, test_device_loop_repeats)
        assert np.array_equal(binfo[i, 0, :, 3:5], sinfo[0, :, 3:5]), (k, binfo[i, 0], sinfo[0])
        assert np.allclose(binfo[i, 0, :, :3], sinfo[0, :, :3], rtol=1e-4, atol=1e-7)
        assert np.abs(binfo[i, :, :, 3:5] - sinfo[:, :, 3:5]).max() <= 3
        for m in range(2):
            dt, dr = pose_diff(bT[i, m], sT[m])
            worst = max(worst, dt, dr)
            assert dt < 1e-4 and dr < 1e-3, (k, m, dt, dr)
        assert np.abs(ba[i] - sa).max() < 1e-3
        worst = max(worst, float(np.abs(ba[i] - sa).max()))
        moved = float(np.abs(bT[i, 0] - T0[i, 0]).max())
        assert moved > 1e-4
    # distinct frames: distinct start poses, observations and results
    assert len({bT[i].tobytes() for i in range(len(ks))}) == len(ks)
    print("%dx%d, %d frames x %d iterations: batch vs single-frame loop, worst pose difference %.2e" % (size[0], size[1], len(ks), iters, worst))


def test_frame_result_does_not_depend_on_the_batch(gpu):
    ks = (0, 1, 2, 3, 4)
    tr, scs, vert, norm, T0, a0 = frames(320, 180, ks, 3)
    bT, ba = tr.optimizePosesBatch(vert, norm, T0, a0, debug=True)
    info = tr.iteration_summaries.copy()
    perm = np.array([3, 0, 4, 2, 1])
    pT, pa = tr.optimizePosesBatch(vert[perm].contiguous(), norm[perm].contiguous(), T0[perm], a0[perm], debug=True)
    pinfo = tr.iteration_summaries.copy()
    assert np.array_equal(pinfo[:, 0, :, 3:5], info[perm][:, 0, :, 3:5])
    assert np.abs(pT - bT[perm]).max() < 2e-5 and np.abs(pa - ba[perm]).max() < 1e-4
    # a batch of one and a chunked batch (DARTK_BATCH_CHUNK is read on every call; the chunk seams are covered by tests/test_gpu_frame_batch_chunks.py)
    oT, oa = tr.optimizePosesBatch(vert[2:3].contiguous(), norm[2:3].contiguous(), T0[2:3], a0[2:3])
    assert np.abs(oT[0] - bT[2]).max() < 2e-5 and np.abs(oa[0] - ba[2]).max() < 1e-4
    # the tracker's own poses are untouched by a batch call
    before = [tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)]
    tr.optimizePosesBatch(vert, norm, T0, a0)
    assert all(np.array_equal(before[m], tr.getPose(m).getTransformCameraToModel()) for m in range(2))


def test_batch_frame_against_the_oracle(gpu):
    """One frame of a batch against the CPU oracle directly (segment by segment, tests/trajectory.py)."""
    from trajectory import parts, segmented_parity
    ks = (2, 5, 7)
    iters = 4
    tr, scs, vert, norm, T0, a0 = frames(320, 180, ks, iters)
    i = 1
    v_np, n_np = vert[i].cpu().numpy(), norm[i].cpu().numpy()

    def gpu_prefix(k):
        tr.getOptions().numIterations = k
        bT, ba = tr.optimizePosesBatch(vert, norm, T0, a0, debug=True)
        return [(bT[i, 0].copy(), ba[i].copy()), (bT[i, 1].copy(), np.zeros(0, dtype=F))], tr.iteration_summaries[i].copy()

    def oracle_from(pose, k0, n):
        tr.getOptions().numIterations = n
        with O.obs_sdf_start(T0[i]):
            oT, oa, oinfo, _ = O.optimize_poses(tr, v_np, n_np, [p[0] for p in pose], [p[1] for p in pose])
        return [(oT[m], oa[m]) for m in range(2)], oinfo

    def pose_err(a, b):
        d = [pose_diff(a[m][0], b[m][0]) for m in range(2)]
        return parts(translation_m=[x[0] for x in d], rotation_rad=[x[1] for x in d],
                     joints_rad=[np.abs(a[m][1] - b[m][1]).max() for m in range(2) if len(b[m][1])])

    segmented_parity("C5 frame %d inside a batch of %d" % (ks[i], len(ks)), iters, gpu_prefix, oracle_from,
                     lambda g, o: bool(np.array_equal(g[:, 3:5], o[:, 3:5])), pose_err, bar=1e-3)
