"""BASELINE configs[4] as it is stated -- a batch of MORE frames than one chunk of dartk_optimize_poses_batch holds (64): the
chunk seam is crossed under a checker. 130 frames of the config-5 sequence at 160x90 (three chunks: 64 + 64 + 2): frames 0, 63,
64, 65 and the last against the single-frame device loop to the bars of test_gpu_frame_batch.py, one frame of the second chunk
against the CPU oracle segment by segment, and the same frames with the seam moved (DARTK_BATCH_CHUNK) against the default
chunking. Reference loop: optimizer.cpp:524-781 (one frame per call there; parity unpinned by the reference's own tests)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from dart_amd import configs
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu
N_FRAMES, W, H, ITERS = 130, 160, 90, 4


@pytest.fixture(scope="module")
def batch(gpu):
    sc = configs.config5(W, H, 0)
    tr = configs.make_multi_tracker(sc)
    tr.getOptions().numIterations = ITERS
    vert, norm, T0, a0 = configs.config5_batch(tr, sc, range(N_FRAMES))
    return tr, sc, vert, norm, T0, a0


class chunk_env:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("DARTK_BATCH_CHUNK")
        os.environ["DARTK_BATCH_CHUNK"] = str(self.value)

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ["DARTK_BATCH_CHUNK"]
        else:
            os.environ["DARTK_BATCH_CHUNK"] = self.old
        return False


def single(tr, T0, a0, vert, norm):
    for m in range(2):
        tr.getPose(m).setTransformCameraToModel(T0[m])
    tr.getPose(0).getReducedArticulation()[:] = a0
    tr.optimizePosesDevice(vert, norm, debug=True)
    return ([tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)], tr.getPose(0).getReducedArticulation().copy(),
            tr.iteration_summaries.copy())


def test_frames_across_the_chunk_seam_equal_the_single_frame_loop(batch):
    tr, sc, vert, norm, T0, a0 = batch
    assert "DARTK_BATCH_CHUNK" not in os.environ         # the default chunk of 64: seams after frames 63 and 127
    tr.getOptions().numIterations = ITERS
    bT, ba = tr.optimizePosesBatch(vert, norm, T0, a0, debug=True)
    binfo = tr.iteration_summaries.copy()
    assert bT.shape[0] == N_FRAMES and np.isfinite(bT).all() and np.isfinite(ba).all() and np.isfinite(binfo).all()
    # every frame did its work: points associated and predicted in every iteration, the pose moved
    assert (binfo[:, :, 0, 3] > 50).all() and (binfo[:, :, 0, 4] > 20).all(), "a frame of the batch was skipped"
    assert (np.abs(bT[:, 0] - T0[:, 0]).reshape(N_FRAMES, -1).max(axis=1) > 1e-4).all()
    assert len({bT[i].tobytes() for i in range(N_FRAMES)}) == N_FRAMES
    worst = 0.0
    for i in (0, 63, 64, 65, 127, 128, N_FRAMES - 1):
        sT, sa, sinfo = single(tr, T0[i], a0[i], vert[i], norm[i])
        assert np.array_equal(binfo[i, 0, :, 3:5], sinfo[0, :, 3:5]), (i, binfo[i, 0], sinfo[0])
        assert np.allclose(binfo[i, 0, :, :3], sinfo[0, :, :3], rtol=1e-4, atol=1e-7)
        assert np.abs(binfo[i, :, :, 3:5] - sinfo[:, :, 3:5]).max() <= 3
        for m in range(2):
            dt, dr = pose_diff(bT[i, m], sT[m])
            worst = max(worst, dt, dr)
            assert dt < 1e-4 and dr < 1e-3, (i, m, dt, dr)
        assert np.abs(ba[i] - sa).max() < 1e-3
        worst = max(worst, float(np.abs(ba[i] - sa).max()))
    print("%d frames (chunks of 64) x %d iterations: frames 0/63/64/65/127/128/%d vs the single-frame loop, worst pose difference %.2e"
          % (N_FRAMES, ITERS, N_FRAMES - 1, worst))


@pytest.mark.parametrize("chunk", [1, 2, 7, 50, 1000, 0, -3, "abc"])
def test_moving_the_seam_changes_nothing(batch, chunk):
    """a frame's result does not depend on the chunk it falls in; 0, a negative number or text fall back to 64 (and terminate)"""
    tr, sc, vert, norm, T0, a0 = batch
    n = 23
    tr.getOptions().numIterations = 3
    v, nm = vert[:n].contiguous(), norm[:n].contiguous()
    rT, ra = tr.optimizePosesBatch(v, nm, T0[:n], a0[:n], debug=True)
    rinfo = tr.iteration_summaries.copy()
    with chunk_env(chunk):
        cT, ca = tr.optimizePosesBatch(v, nm, T0[:n], a0[:n], debug=True)
    cinfo = tr.iteration_summaries.copy()
    assert np.array_equal(cinfo[:, 0, :, 3:5], rinfo[:, 0, :, 3:5])
    assert np.abs(cT - rT).max() < 2e-5 and np.abs(ca - ra).max() < 1e-4


def test_frame_of_the_second_chunk_against_the_oracle(batch):
    from trajectory import parts, segmented_parity
    tr, sc, vert, norm, T0, a0 = batch
    i = 70
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

    segmented_parity("C5 frame %d (second chunk) of a batch of %d" % (i, N_FRAMES), ITERS, gpu_prefix, oracle_from,
                     lambda g, o: bool(np.array_equal(g[:, 3:5], o[:, 3:5])), pose_err, bar=1e-3)
