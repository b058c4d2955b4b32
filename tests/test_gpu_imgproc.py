"""depth -> vertices -> normals (img_proc/organized_point_cloud.cu:11-40, :265-305): the HIP kernels through the
C ABI against the oracle's restatement, BIT FOR BIT, on a map with 20 % holes and on a ray-cast hand; and the oracle against
an independent numpy construction (CPU suite).

Parity unpinned by the reference: it holds no test or fixture for these two kernels; its device build used
--use_fast_math, whose approximate division / rsqrt this path (IEEE division and square root on both sides) does not restate."""
import numpy as np
import pytest

import oracle_lib as O

F = np.float32


def holes_depth(w, h, seed, p_hole=0.2):
    """depth with 20 % zeros, some values below / above / exactly on both edges of the range [0.4, 1.4]"""
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.3, 1.5, size=(h, w)).astype(F)
    d[rng.random((h, w)) < p_hole] = 0
    d[0, 0], d[0, 1], d[1, 0], d[1, 1] = F(0.4), F(1.4), np.nextafter(F(0.4), F(0)), np.nextafter(F(1.4), F(2))
    d[h - 1, w - 1] = F(0.4)
    return d


def numpy_vertices(depth, pp, fl, rng_):
    h, w = depth.shape
    u, v = np.meshgrid(np.arange(w, dtype=F), np.arange(h, dtype=F))
    ok = (depth >= F(rng_[0])) & (depth <= F(rng_[1]))
    out = np.zeros((h, w, 4), dtype=F)
    out[..., 0] = np.where(ok, (u - F(pp[0])) * (depth / F(fl[0])), 0)
    out[..., 1] = np.where(ok, (v - F(pp[1])) * (depth / F(fl[1])), 0)
    out[..., 2] = np.where(ok, depth, 0)
    out[..., 3] = ok
    return out


def numpy_normals(vert):
    """independent construction: padded shifts instead of per-pixel neighbour selection"""
    h, w = vert.shape[:2]
    valid = vert[..., 3] != 0

    def nb(dy, dx):
        ys = np.clip(np.arange(h) + dy, 0, h - 1)
        xs = np.clip(np.arange(w) + dx, 0, w - 1)
        n = vert[ys][:, xs]
        return np.where((n[..., 3] != 0)[..., None], n, vert)[..., :3]
    vX = nb(0, 1) - nb(0, -1)
    vY = nb(1, 0) - nb(-1, 0)
    n = np.stack([vY[..., 1] * vX[..., 2] - vY[..., 2] * vX[..., 1], vY[..., 2] * vX[..., 0] - vY[..., 0] * vX[..., 2],
                  vY[..., 0] * vX[..., 1] - vY[..., 1] * vX[..., 0]], axis=-1).astype(F)
    len2 = (n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1] + n[..., 2] * n[..., 2]).astype(F)
    good = valid & (len2 > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F(1) / np.sqrt(len2, dtype=F)).astype(F)
        scaled = n * inv[..., None]
    out = np.zeros((h, w, 4), dtype=F)
    out[..., :3] = np.where(good[..., None], scaled, 0)
    out[..., 3] = good
    return out


def test_oracle_depth_front_end_matches_independent_numpy():
    w, h = 47, 33
    pp, fl, rg = (23.5, 16.0), (100.0, 110.0), (0.4, 1.4)
    d = holes_depth(w, h, 1)
    v = O.depth_to_vertices(d, pp, fl, rg)
    assert np.array_equal(v.view(np.int32), numpy_vertices(d, pp, fl, rg).view(np.int32))
    # both edges of the range are inside, one ulp beyond is outside (organized_point_cloud.cu:28)
    assert v[0, 0, 3] == 1 and v[0, 1, 3] == 1 and v[1, 0, 3] == 0 and v[1, 1, 3] == 0
    # an out-of-range depth only clears w: x, y, z keep what the map held (organized_point_cloud.cu:36-38)
    prev = np.full((h, w, 4), 7.0, dtype=F)
    v2 = O.depth_to_vertices(d, pp, fl, rg, out=prev.copy())
    bad = v2[..., 3] == 0
    assert bad.sum() > 100 and (v2[bad][:, :3] == 7.0).all() and np.array_equal(v2[~bad], v[~bad])
    n = O.vertices_to_normals(v)
    assert np.array_equal(n.view(np.int32), numpy_normals(v).view(np.int32))
    ok = n[..., 3] == 1
    assert ok.sum() > 0.4 * w * h and np.abs(np.linalg.norm(n[ok][:, :3], axis=1) - 1).max() < 1e-6
    assert (n[v[..., 3] == 0] == 0).all()
    # len2 == 0: an isolated valid pixel (all four neighbours invalid) has vX = vY = 0 -> (0,0,0,0)
    iso = np.zeros((5, 5, 4), dtype=F)
    iso[2, 2] = (0.1, 0.2, 0.9, 1)
    assert (O.vertices_to_normals(iso) == 0).all()
    # a plane facing the camera: cross(vY, vX) points along -z (the reference's sign, organized_point_cloud.cu:291-293)
    flat = O.depth_to_vertices(np.full((6, 8), 0.7, dtype=F), (4, 3), (50.0, 50.0), (0.1, 5.0))
    assert np.allclose(O.vertices_to_normals(flat)[..., :3], (0, 0, -1))
    # 1 x 1 and single-row / single-column images: every neighbour is the pixel itself
    assert (O.vertices_to_normals(flat[:1, :1].copy()) == 0).all()
    assert (O.vertices_to_normals(flat[:1].copy()) == 0).all() and (O.vertices_to_normals(flat[:, :1].copy()) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(80, 60), (641, 479), (1, 1), (64, 1), (1, 37)])
def test_depth_to_vertices_and_normals_bit_exact_with_holes(gpu, size):
    import torch
    from dart_amd.optimizer import depth_to_vertices, vertices_to_normals
    w, h = size
    pp, fl, rg = (w / 2 - 0.5, h / 2.0), (537.0, 541.5), (0.4, 1.4)
    d = holes_depth(w, h, 5) if w > 1 and h > 1 else np.random.default_rng(2).uniform(0.3, 1.5, size=(h, w)).astype(F)
    dd = torch.from_numpy(d).to(gpu)
    ref_v = O.depth_to_vertices(d, pp, fl, rg)
    got_v = depth_to_vertices(dd, pp, fl, rg)
    assert np.array_equal(got_v.cpu().numpy().view(np.int32), ref_v.view(np.int32))
    # only w is cleared where the depth is out of range
    prev = np.random.default_rng(9).normal(size=(h, w, 4)).astype(F)
    got2 = depth_to_vertices(dd, pp, fl, rg, out=torch.from_numpy(prev.copy()).to(gpu)).cpu().numpy()
    assert np.array_equal(got2.view(np.int32), O.depth_to_vertices(d, pp, fl, rg, out=prev.copy()).view(np.int32))
    ref_n = O.vertices_to_normals(ref_v)
    got_n = vertices_to_normals(got_v).cpu().numpy()
    assert np.array_equal(got_n.view(np.int32), ref_n.view(np.int32))
    if w > 8 and h > 8:
        assert 0.35 < (ref_n[..., 3] == 1).mean() < 0.9 and (ref_v[..., 3] == 0).mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["allegro", "left_hand"])
def test_front_end_bit_exact_on_raycast_hand(gpu, name):
    """the synthetic-observation generator's own input: the ray-cast depth of a posed hand at 640x480 (silhouette edges,
    self-occlusion steps, background zeros)"""
    import scenes
    from dart_amd import device as D, se3
    from dart_amd.optimizer import synthetic_observation
    m = scenes.articulated(name)
    dm = D.DeviceModel(m, model_id=0)
    T_cm = scenes.random_T(61, trans=0.02, rot=0.4, base=(0, 0, 0.45))
    dm.set_pose(se3.invert(T_cm), scenes.random_joints(m, 4))
    w, h, fl, pp = 640, 480, (537.0, 537.0), (320.0, 240.0)
    depth, vert, norm = synthetic_observation([dm], w, h, fl, pp)
    d = depth.cpu().numpy()
    assert 5000 < (d > 0).sum() < 0.5 * w * h
    ref_v = O.depth_to_vertices(d, pp, fl)
    assert np.array_equal(vert.cpu().numpy().view(np.int32), ref_v.view(np.int32))
    ref_n = O.vertices_to_normals(ref_v)
    assert np.array_equal(norm.cpu().numpy().view(np.int32), ref_n.view(np.int32))
    assert (ref_n[..., 3] == 1).sum() > 5000
