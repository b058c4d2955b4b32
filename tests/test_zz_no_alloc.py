"""dartk_ne_obs2mod_noalloc (the entry point that allocates nothing: its partial-system scratch is a fixed device array inside
the library) against dartk_ne_obs2mod, through the C ABI, on two calls in a row, the second with a smaller input. Bar: the
packed systems agree within 1e-4 norm-relative per block (north_star's J^T J / J^T e tolerance) -- and bit for bit when both
entry points cut the list into the same workgroups (<= 256 tiles of 128 points)."""
import numpy as np
import pytest
import torch

import scenes
from dart_amd import device as D
from dart_amd import se3
from dart_amd._abi import lib
from dart_amd.device import _ptr, _stream

F = np.float32
pytestmark = pytest.mark.gpu


def rel_err_blocks(got, ref, dims):
    out = 0.0
    for sl in (slice(0, dims), slice(dims, len(ref) - 1), slice(len(ref) - 1, len(ref))):
        n = np.linalg.norm(ref[sl].astype(np.float64))
        d = np.linalg.norm(got[sl].astype(np.float64) - ref[sl].astype(np.float64))
        out = max(out, d / n if n > 0 else d)
    return out


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_no_alloc_entry_matches_allocating_entry(gpu):
    m = scenes.articulated("allegro")
    dm = D.DeviceModel(m, model_id=0)
    joints = scenes.random_joints(m, 11)
    T_mc = scenes.random_T(21, base=(0, 0, -0.5))
    dm.set_pose(T_mc, joints)
    _, T_mf, _, _ = dm.get_transforms()
    # first call: 320x240 (> 256 tiles of associated points: the fixed scratch forces a strided grid); second: a smaller map
    results = []
    for (w, h, seed) in ((320, 240, 3), (96, 64, 4)):
        vert = scenes.points_near_model(m, T_mf, se3.invert(T_mc), w, h, seed)
        v_d = torch.from_numpy(vert).to(gpu)
        pts, count, _ = D.error_and_data_association(v_d, torch.zeros_like(v_d), dm, dist_threshold=0.05)
        assert count > 1000
        out = torch.full((D.sys_len(m.dims),), float("nan"), dtype=torch.float32, device=gpu)
        ref = D.norm_eqns_obs_to_mod(v_d, dm, pts, count, huber_delta=0.02).cpu().numpy()
        before = free_bytes()
        got_d = D.norm_eqns_obs_to_mod(v_d, dm, pts, count, huber_delta=0.02, result=out, no_alloc=True)
        assert free_bytes() == before, "the no-allocation entry point changed the device's free memory"
        got = got_d.cpu().numpy()
        tiles = (count + 127) // 128
        err = rel_err_blocks(got, ref, m.dims)
        print("%dx%d: %d points, %d tiles: no-alloc vs allocating entry %.2e%s" % (w, h, count, tiles, err, " (bit-exact)" if np.array_equal(got.view(np.int32), ref.view(np.int32)) else ""))
        assert np.isfinite(got).all()
        assert err < 1e-4
        if tiles <= 256:
            assert np.array_equal(got.view(np.int32), ref.view(np.int32))
        results.append((tiles, got))
    assert results[0][0] > 256 >= results[1][0], "the first call must use more slabs than the second (stale slabs must not leak)"
    # the <= 10 elements gate (obsToMod.cu:644-646) and the size limit
    out = torch.full((D.sys_len(m.dims),), float("nan"), dtype=torch.float32, device=gpu)
    z = D.norm_eqns_obs_to_mod(v_d, dm, pts, 7, result=out, no_alloc=True).cpu().numpy()
    assert not z.any()
    # a system beyond the fixed array's 32 dimensions is refused with a clear error 
    rc = lib().dartk_ne_obs2mod_noalloc(40, _ptr(v_d), 96, 64, dm.handle, 0.02, _ptr(pts), count, _ptr(out), _stream())
    assert rc == -1 and b"at most 32 dimensions" in lib().dartk_last_error()


def test_no_alloc_entry_small_system_uses_the_same_kernel(gpu):
    """A packed system of at most 128 entries (dims <= 14): dartk_ne_obs2mod runs the one-output-per-thread instantiation, and so
    does the no-allocation entry -- bit-identical while both cut the list into the same workgroups. The rigid sphere (6 dims,
    28 entries) and Barrett (6 + 8 dims, 120 entries)."""
    for name, n_min in (("sphere", 300), ("barrett", 300)):
        m = scenes.sphere() if name == "sphere" else scenes.articulated(name)
        dm = D.DeviceModel(m, model_id=0)
        joints = scenes.random_joints(m, 5) if m.n_joints else np.zeros(0, dtype=F)
        T_mc = scenes.random_T(22, base=(0, 0, -0.5))
        dm.set_pose(T_mc, joints)
        _, T_mf, _, _ = dm.get_transforms()
        vert = scenes.points_near_model(m, T_mf, se3.invert(T_mc), 128, 96, 8)
        v_d = torch.from_numpy(vert).to(gpu)
        pts, count, _ = D.error_and_data_association(v_d, torch.zeros_like(v_d), dm, dist_threshold=0.05)
        assert count > n_min and (count + 127) // 128 <= 256
        ref = D.norm_eqns_obs_to_mod(v_d, dm, pts, count, huber_delta=0.02).cpu().numpy()
        out = torch.full((D.sys_len(m.dims),), float("nan"), dtype=torch.float32, device=gpu)
        before = free_bytes()
        got_d = D.norm_eqns_obs_to_mod(v_d, dm, pts, count, huber_delta=0.02, result=out, no_alloc=True)
        assert free_bytes() == before
        got = got_d.cpu().numpy()
        assert np.isfinite(got).all() and np.abs(ref).max() > 0
        assert np.array_equal(got.view(np.int32), ref.view(np.int32)), (name, m.dims, rel_err_blocks(got, ref, m.dims))
