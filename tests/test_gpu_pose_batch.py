"""GPU parity: dartk_optimize_pose_batch (Optimizer::optimizePose, optimizer.cpp:423-522, for independent frames side by side) --
every frame of the batch must equal the single-frame call on that frame BIT FOR BIT (same kernels, same partial systems, same
order of every sum; no atomically appended list on this path), whatever the chunking; one frame is also held against the CPU
oracle to the single call's bars (first-iteration counts equal, poses within 1e-3 m / 1e-3 rad)."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dart_amd import configs
from dart_amd import device as D
from dart_amd.optimizer import Optimizer, synthetic_observation
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu


def make_frames(sc, n):
    """observation maps of n frames rendered from their true poses + the start poses"""
    tT, tj, iT, ij = configs.tracking_frames(sc, n)
    dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
    verts, norms = [], []
    for f in range(n):
        dm.set_pose(tT[f], tj[f])
        _, v, nm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
        verts.append(v); norms.append(nm)
    return dm, torch.stack(verts).contiguous(), torch.stack(norms).contiguous(), iT, ij


def singles(dm, opt, sc, vert, norm, iT, ij, **kw):
    out = []
    for f in range(len(iT)):
        dm.set_pose(iT[f], ij[f])
        opt.optimizePose(dm, vert[f], norm[f], sc.opts, debug=True, **kw)
        T, _, _, j = dm.get_transforms()
        out.append((T.copy(), j.copy(), opt.iter_info.cpu().numpy().copy()))
    return out


def batch(dm, opt, sc, vert, norm, iT, ij, **kw):
    T = torch.from_numpy(iT.copy()).cuda()
    j = torch.from_numpy(ij.copy()).cuda()
    opt.optimizePoseBatch(dm, vert, norm, T, j, sc.opts, debug=True, **kw)
    return T.cpu().numpy(), j.cpu().numpy(), opt.iter_info.cpu().numpy()


def assert_bitwise(single, bT, bj, binfo, iters):
    for f, (T, j, info) in enumerate(single):
        assert T.astype(F).tobytes() == bT[f].tobytes(), ("T_mc", f, np.abs(T - bT[f]).max())
        assert j.astype(F).tobytes() == bj[f].tobytes(), ("joints", f)
        assert info[:iters].tobytes() == binfo[f, :iters].tobytes(), ("iteration records", f)


@pytest.mark.parametrize("chunk_env", [None, "2", "1"])
def test_batch_equals_single_calls_bitwise(gpu, chunk_env, monkeypatch):
    sc = configs.config2(320, 240)
    sc.opts.numIterations = 4
    n = 5
    dm, vert, norm, iT, ij = make_frames(sc, n)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    single = singles(dm, opt, sc, vert, norm, iT, ij)
    pose_before = dm.get_transforms()
    if chunk_env is not None:
        monkeypatch.setenv("DARTK_BATCH_CHUNK", chunk_env)
    bT, bj, binfo = batch(dm, opt, sc, vert, norm, iT, ij)
    assert_bitwise(single, bT, bj, binfo, 4)
    # the frames are different problems, and each one moved
    assert len({bT[f].tobytes() for f in range(n)}) == n
    assert all(np.abs(bT[f] - iT[f]).max() > 1e-5 for f in range(n))
    assert all(binfo[f, 0, 3] > 1000 and binfo[f, 0, 4] > 200 for f in range(n))
    # the handle's own pose is not touched by the batch
    after = dm.get_transforms()
    assert all(np.array_equal(a, b) for a, b in zip(pose_before, after))


def test_batch_with_self_intersection_and_term_switches(gpu):
    sc = configs.config2(320, 240)
    sc.opts.numIterations = 3
    n = 3
    dm, vert, norm, iT, ij = make_frames(sc, n)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    m = sc.host_model
    cloud = torch.from_numpy(m.collision_cloud(2e4, seed=5)).cuda()
    ns = len(m.sdfs)
    pot = torch.from_numpy((1 - np.eye(ns, dtype=np.int32)).reshape(-1).copy()).cuda()
    sc.opts.lambdaIntersection = [2.0]
    kw = dict(collision_cloud=cloud, potential=pot)
    single = singles(dm, opt, sc, vert, norm, iT, ij, **kw)
    bT, bj, binfo = batch(dm, opt, sc, vert, norm, iT, ij, **kw)
    assert_bitwise(single, bT, bj, binfo, 3)
    assert any(binfo[f, 0, 2] > 0 for f in range(n)), "the intersection term is active in none of the frames"
    # obs->mod only / mod->obs only
    sc.opts.lambdaIntersection = [0.0]
    for lam_a, lam_b in ((1.0, 0.0), (0.0, 1.0)):
        sc.opts.lambdaObsToMod, sc.opts.lambdaModToObs = lam_a, lam_b
        single = singles(dm, opt, sc, vert, norm, iT, ij)
        bT, bj, binfo = batch(dm, opt, sc, vert, norm, iT, ij)
        assert_bitwise(single, bT, bj, binfo, 3)
        assert np.abs(bT - iT).max() > 1e-5


def test_batch_rigid_model_and_zero_iterations(gpu):
    sc = configs.config1(160, 120)
    sc.opts.numIterations = 3
    n = 4
    dm, vert, norm, iT, ij = make_frames(sc, n)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    single = singles(dm, opt, sc, vert, norm, iT, ij)
    bT, bj, binfo = batch(dm, opt, sc, vert, norm, iT, ij)
    assert_bitwise(single, bT, bj, binfo, 3)
    sc.opts.numIterations = 0
    bT, bj, _ = batch(dm, opt, sc, vert, norm, iT, ij)
    assert np.array_equal(bT, iT)


def test_batch_frame_matches_oracle(gpu):
    sc = configs.config2(320, 240)
    sc.opts.numIterations = 3
    n = 4
    dm, vert, norm, iT, ij = make_frames(sc, n)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    bT, bj, binfo = batch(dm, opt, sc, vert, norm, iT, ij)
    f = 2
    om = O.OModel(sc.host_model, 0, sc.obs_layout)
    o_struct = sc.opts.c_struct(0, opt.pred_width, opt.pred_height, opt.pred_fl)
    o_T, o_j, o_info, _ = O.optimize_pose(om, iT[f], ij[f], vert[f].cpu().numpy(), norm[f].cpu().numpy(), o_struct)
    assert binfo[f, 0, 3] == o_info[0, 3] and binfo[f, 0, 4] == o_info[0, 4]
    assert np.all(np.abs(binfo[f, 0, :2] - o_info[0, :2]) <= 2e-5 * np.abs(o_info[0, :2]) + 1e-9)
    dt, dr = pose_diff(bT[f], o_T)
    dj = float(np.abs(bj[f] - o_j).max())
    print("batch frame %d vs oracle: |dt|=%.3e m |dr|=%.3e rad max|dq|=%.3e rad" % (f, dt, dr, dj))
    assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3


def test_batch_rejects_bad_arguments(gpu):
    from dart_amd._abi import DartkError
    sc = configs.config1(160, 120)
    dm, vert, norm, iT, ij = make_frames(sc, 2)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    sc.opts.numIterations = -1
    with pytest.raises(DartkError):
        batch(dm, opt, sc, vert, norm, iT, ij)
