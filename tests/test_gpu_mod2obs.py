"""GPU parity: ray-cast prediction, cull, observation-SDF construction (splat + distance transform) and the
model-to-observation normal equations, through the C ABI, vs the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import scenes
from dart_amd import device as D
from dart_amd import se3
from test_gpu_obs2mod import rel_err_blocks

F = np.float32
pytestmark = pytest.mark.gpu


def posed(name, seed, gpu, obs=True, z=0.45):
    m = scenes.articulated(name) if name != "sphere" else scenes.sphere()
    layout = m.obs_sdf_layout() if obs else None
    om = O.OModel(m, 0, layout)
    dm = D.DeviceModel(m, model_id=0, obs_layout=layout)
    joints = scenes.random_joints(m, seed) if m.n_joints else np.zeros(0, dtype=F)
    T_cm = scenes.random_T(seed + 50, trans=0.02, rot=0.4, base=(0, 0, z))
    T_mc = se3.invert(T_cm)
    dm.set_pose(T_mc, joints)
    g_mc, T_mf, T_fm, _ = dm.get_transforms()
    return m, om, dm, T_mc, T_mf, T_fm, layout


def compare_predictions(got, ref):
    """Bit-exact except rays whose zero crossing sits on a comparison edge: documented tolerance = identical
    hit mask and labels, coordinates equal to 1e-6 m on > 99.9 % of pixels and all bit-equal otherwise."""
    same = got.view(np.int32) == ref.view(np.int32)
    frac = same.all(axis=-1).mean()
    return frac


@pytest.mark.parametrize("name,size", [("sphere", (160, 120)), ("allegro", (160, 120)), ("left_hand", (320, 240))])
def test_raycast_prediction_bitexact(gpu, name, size):
    w, h = size
    m, om, dm, T_mc, T_mf, T_fm, _ = posed(name, 3, gpu, obs=False)
    fl = (537.0 * w / 640, 537.0 * w / 640)
    pp = (w / 2, h / 2)
    pred = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    ref = O.raycast_prediction([om], [(T_mc, T_fm, T_mf)], fl, pp, w, h)
    assert (ref[..., 2] != 0).sum() > 300
    assert np.array_equal(pred.view(np.int32), ref.view(np.int32))
    if name != "sphere":
        labels = np.unique(ref[..., 3][ref[..., 2] != 0])
        assert len(labels) >= 4


def test_cull_unobservable(gpu):
    w, h = 160, 120
    m, om, dm, T_mc, T_mf, T_fm, _ = posed("allegro", 4, gpu, obs=False)
    fl, pp = (537.0 / 4, 537.0 / 4), (w / 2, h / 2)
    pred = D.raycast_prediction([dm], fl, pp, w, h)
    obs = scenes.random_vertex_map(2 * w, 2 * h, 9, p_valid=0.9)
    ref = O.cull_unobservable(pred.cpu().numpy(), obs)
    got = D.cull_unobservable(pred, torch.from_numpy(obs).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))
    assert (got[..., 2] == 0).sum() > (ref[..., 3] < 0).sum()


@pytest.mark.parametrize("shape", [(64, 64, 64), (40, 24, 16), (7, 5, 3), (1, 1, 1), (130, 9, 70)])
def test_distance_transform_matches_envelope_scan(gpu, shape):
    d, h, w = shape
    rng = np.random.default_rng(d * 100 + h)
    # the value classes the path produces: obs-SDF splat {0,500,1000} and occupancy {0,1e20}
    a = rng.choice(np.array([0, 500, 1000], dtype=F), size=shape, p=[0.05, 0.45, 0.5]).astype(F)
    occ = np.where(rng.random(shape) < 0.02, F(0), F(1e20)).astype(F)
    for grid in (a, occ):
        for sq in (True, False):
            ref = O.distance_transform_3d(grid, take_sqrt=sq)
            got = D.distance_transform_3d(torch.from_numpy(grid.copy()).to(gpu), take_sqrt=sq).cpu().numpy()
            assert np.array_equal(got.view(np.int32), ref.view(np.int32))


def test_signed_distance_transform(gpu):
    m = scenes.articulated("left_hand")
    grid, occ = m._occupancy(m.sdf_frame[0], 0.005, 0.1)
    inp = np.where(occ, F(0), F(1e20)).astype(F)
    ref = O.distance_transform_3d(inp, signed=True)
    got = D.distance_transform_3d(torch.from_numpy(inp.copy()).to(gpu), signed=True).cpu().numpy()
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))
    # the host voxeliser (scipy EDT) agrees with both
    np.testing.assert_array_equal(m.sdfs[0].data.view(np.int32), ref.view(np.int32))
    assert (ref < 0).sum() > 10 and (ref == 0).sum() > 10


@pytest.mark.parametrize("name", ["sphere", "allegro"])
def test_obs_sdf_build(gpu, name):
    w, h = 320, 240
    m, om, dm, T_mc, T_mf, T_fm, layout = posed(name, 5, gpu, obs=True, z=0.5)
    fl, pp = (537.0 / 2, 537.0 / 2), (w / 2, h / 2)
    # observation = ray-cast of the posed model; w carries the label, make it a valid vertex map (w=1, misses w=0)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    D.generate_obs_sdf(torch.from_numpy(obs).to(gpu), dm, focal_length=fl[0])
    got = D.obs_sdf_numpy(dm)
    splat = O.splat_obs_sdf(obs, se3.invert(T_mc), layout, focal_length=fl[0])
    assert set(np.unique(splat)) <= {0.0, 500.0, 1000.0} and (splat == 0).sum() > 50
    ref = O.distance_transform_3d(splat, take_sqrt=True)
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))


def test_splat_obs_sdf_zeros_alone_with_its_own_transform(gpu):
    """splatObsSdfZeros as the reference declares it (modToObs.h:68-74): T_cm is an argument; the model's pose stays. Followed by
    the caller's distanceTransform3D (optimizer.cpp:205-215) it gives what dartk_obs_sdf_build gives at that pose."""
    from dart_amd._abi import check, lib, fptr
    w, h = 320, 240
    m, om, dm, T_mc, T_mf, T_fm, layout = posed("allegro", 6, gpu, obs=True, z=0.5)
    fl, pp = (537.0 / 2, 537.0 / 2), (w / 2, h / 2)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    obs_d = torch.from_numpy(obs).to(gpu)
    T_other = se3.mul(se3.invert(T_mc), se3.exp(np.array([0.01, -0.02, 0.015, 0.05, -0.03, 0.04], dtype=F)))     # not the model's T_cm
    before = dm.get_transforms()
    check(lib().dartk_splat_obs_sdf_zeros(D._ptr(obs_d), w, h, fptr(np.ascontiguousarray(T_other, dtype=F)), dm.handle, fl[0], D._stream()))
    got = D.obs_sdf_numpy(dm)
    ref = O.splat_obs_sdf(obs, T_other, layout, focal_length=fl[0])
    assert np.array_equal(got.view(np.int32), ref.view(np.int32)) and (ref == 0).sum() > 50
    assert not np.array_equal(ref, O.splat_obs_sdf(obs, se3.invert(T_mc), layout, focal_length=fl[0]))
    assert all(np.array_equal(a, b) for a, b in zip(before, dm.get_transforms()))
    # the caller's next two steps: distance transform into a scratch volume, copy back
    n = got.size
    scratch = torch.empty(n, dtype=torch.float32, device=gpu)
    ptr = lib().dartk_model_obs_sdf_data(dm.handle)
    import ctypes as C
    check(lib().dartk_distance_transform_3d(C.c_void_p(ptr), D._ptr(scratch), layout[0][0], layout[0][1], layout[0][2], 1, D._stream()))
    dt_ref = O.distance_transform_3d(ref, take_sqrt=True)
    assert np.array_equal(scratch.cpu().numpy().reshape(dt_ref.shape).view(np.int32), dt_ref.view(np.int32))


@pytest.mark.parametrize("name", ["allegro", "left_hand"])
def test_mod2obs_normal_equations(gpu, name):
    w, h = 160, 120
    m, om, dm, T_mc, T_mf, T_fm, layout = posed(name, 6, gpu, obs=True, z=0.5)
    fl, pp = (537.0 / 4, 537.0 / 4), (w / 2, h / 2)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    D.generate_obs_sdf(torch.from_numpy(obs).to(gpu), dm, focal_length=fl[0])
    obs_data = D.obs_sdf_numpy(dm)
    T_gc = T_mc.copy()                                   # obs-SDF frame = model frame at build time (optimizer.cpp:448)
    # perturb the pose, predict, and evaluate the term
    joints2 = scenes.random_joints(m, 77)
    T_mc2 = se3.mul(se3.exp(np.array([0.01, -0.008, 0.005, 0.03, -0.02, 0.04], dtype=F)), T_mc)
    dm.set_pose(T_mc2, joints2)
    _, T_mf2, T_fm2, _ = dm.get_transforms()
    pred = D.raycast_prediction([dm], fl, pp, w, h)
    og = O.make_grid(layout[0], layout[1], layout[2], obs_data)
    res, cnt = D.norm_eqns_mod_to_obs(pred, dm, T_gc=T_gc)
    ref, n = O.ne_mod2obs(pred.cpu().numpy(), om, T_gc, T_fm2, T_mf2, og)
    assert n > 300 and int(cnt.item()) == n
    assert rel_err_blocks(res.cpu().numpy(), ref, m.dims) < 1e-4
    # truncated variant uses the model's current T_mc
    res, cnt = D.norm_eqns_mod_to_obs(pred, dm, truncation=0.02)
    ref, n = O.ne_mod2obs(pred.cpu().numpy(), om, T_mc2, T_fm2, T_mf2, og, truncation=0.02)
    assert int(cnt.item()) == n and n > 20
    assert rel_err_blocks(res.cpu().numpy(), ref, m.dims) < 1e-4
    # error only
    e = D.error_mod_to_obs(pred, dm)
    e_ref = O.error_mod2obs(pred.cpu().numpy(), og)
    assert abs(e - e_ref) <= 1e-5 * max(abs(e_ref), 1e-12)
    # wrong model label / empty prediction => zero system, zero count
    empty = torch.zeros_like(pred)
    res, cnt = D.norm_eqns_mod_to_obs(empty, dm, T_gc=T_gc)
    assert int(cnt.item()) == 0 and not res.cpu().numpy().any()


def test_mod2obs_reduced_variants(gpu):
    w, h = 160, 120
    m, om, dm, T_mc, T_mf, T_fm, layout = posed("barrett", 8, gpu, obs=True, z=0.5)
    fl, pp = (537.0 / 4, 537.0 / 4), (w / 2, h / 2)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    D.generate_obs_sdf(torch.from_numpy(obs).to(gpu), dm, focal_length=fl[0])
    og = O.make_grid(layout[0], layout[1], layout[2], D.obs_sdf_numpy(dm))
    T_mc2 = se3.mul(se3.exp(np.array([0.01, -0.008, 0.005, 0.03, -0.02, 0.04], dtype=F)), T_mc)
    dm.set_pose(T_mc2, scenes.random_joints(m, 78))
    _, T_mf2, T_fm2, _ = dm.get_transforms()
    pred = D.raycast_prediction([dm], fl, pp, w, h)
    A = np.zeros((8, 4), dtype=F)
    for f_, r, c in [(0, 0, 1), (3, 0, 1), (1, 1, 1), (2, 1, 1 / 3), (4, 2, 1), (5, 2, 1 / 3), (6, 3, 1), (7, 3, 1 / 3)]:
        A[f_, r] = c
    counter = torch.full((1,), 7, dtype=torch.int32, device=gpu)        # reduced wrappers do NOT zero it (A-19)
    res, cnt = D.norm_eqns_mod_to_obs(pred, dm, reduction=("linear", torch.from_numpy(A.reshape(-1)).to(gpu), 10), counter=counter)
    ref, n = O.ne_mod2obs(pred.cpu().numpy(), om, T_mc2, T_fm2, T_mf2, og, reduction=("linear", A.reshape(-1), 10))
    assert int(cnt.item()) == n + 7 and n > 100
    assert rel_err_blocks(res.cpu().numpy(), ref, 10) < 1e-4
    mapping = np.array([0, 1, 1, 0, 2, 2, 3, 3], dtype=np.int32)
    res, cnt = D.norm_eqns_mod_to_obs(pred, dm, reduction=("map", torch.from_numpy(mapping).to(gpu), 10))
    ref, n = O.ne_mod2obs(pred.cpu().numpy(), om, T_mc2, T_fm2, T_mf2, og, reduction=("map", mapping, 10))
    assert rel_err_blocks(res.cpu().numpy(), ref, 10) < 1e-4
