"""GPU parity: the kernels' optional debug outputs -- the mod->obs debug maps (debugDataAssociation / debugError / debugNorm,
modToObs.cu:47-82,145-186; optimizer.cpp:296-324) and the intersection terms' debugError (intersection.cu:131-133,303-313,
433-435; optimizer.cpp:646-682) -- vs the oracle's restatement. Values are index / gather work: bit-exact. The reference's tests
never request them ("parity unpinned")."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import scenes
from dart_amd import device as D
from dart_amd import se3
from test_gpu_mod2obs import posed
from test_gpu_intersection import closed_fist, potential_matrix

F = np.float32
pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int32), np.ascontiguousarray(b).view(np.int32))


@pytest.mark.parametrize("name", ["allegro", "left_hand"])
def test_mod2obs_debug_maps(gpu, name):
    w, h = 160, 120
    m, om, dm, T_mc, T_mf, T_fm, layout = posed(name, 6, gpu, obs=True, z=0.5)
    fl, pp = (537.0 / 4, 537.0 / 4), (w / 2, h / 2)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    D.generate_obs_sdf(torch.from_numpy(obs).to(gpu), dm, focal_length=fl[0])
    og = O.make_grid(layout[0], layout[1], layout[2], D.obs_sdf_numpy(dm))
    T_gc = T_mc.copy()
    T_mc2 = se3.mul(se3.exp(np.array([0.01, -0.008, 0.005, 0.03, -0.02, 0.04], dtype=F)), T_mc)
    dm.set_pose(T_mc2, scenes.random_joints(m, 77))
    pred = D.raycast_prediction([dm], fl, pp, w, h)
    p_np = pred.cpu().numpy()
    # plain kernel (given T_gc) and truncated kernel (the model's own transform), all three maps
    for trunc, T in ((None, T_gc), (0.02, None)):
        da, err, norm = D.mod_to_obs_debug_maps(pred, dm, T_gc=T, truncation=trunc)
        rda, rerr, rnorm = O.mod2obs_debug_maps(p_np, 0, T_gc if T is not None else T_mc2, og, truncation=trunc)
        assert np.array_equal(da.cpu().numpy(), rda)
        assert same_bits(err.cpu().numpy(), rerr)                 # NaN where nothing was written: compared as bit patterns
        assert same_bits(norm.cpu().numpy(), rnorm)
        n_lab = int((rda >= 0).sum())
        assert n_lab > (300 if trunc is None else 20) and np.isnan(rerr).sum() > 1000 and (rnorm[:, 3] == 1).sum() == n_lab
        # the labelled pixels are exactly the ones the normal equations count
        _, cnt = D.norm_eqns_mod_to_obs(pred, dm, T_gc=T, truncation=trunc) if trunc is None else D.norm_eqns_mod_to_obs(pred, dm, truncation=trunc)
        assert int(cnt.item()) == n_lab
    # a single map on request; the others are left alone
    only = D.mod_to_obs_debug_maps(pred, dm, T_gc=T_gc)[1]
    assert same_bits(only.cpu().numpy(), O.mod2obs_debug_maps(p_np, 0, T_gc, og)[1])


def test_mod2obs_debug_maps_second_model_keeps_the_first_models_entries(gpu):
    """modelNum != 0 does not clear the label and residual maps (modToObs.cu:47-48): the maps of a frame are filled model after
    model. A model with ID 1 on the same prediction: no label of its own, model 0's entries stay."""
    w, h = 160, 120
    m, om, dm, T_mc, T_mf, T_fm, layout = posed("allegro", 6, gpu, obs=True, z=0.5)
    dm1 = D.DeviceModel(m, model_id=1, obs_layout=layout)
    fl, pp = (537.0 / 4, 537.0 / 4), (w / 2, h / 2)
    obs = D.raycast_prediction([dm], fl, pp, w, h).cpu().numpy()
    obs[..., 3] = (obs[..., 2] != 0).astype(F)
    obs[obs[..., 3] == 0] = 0
    for d in (dm, dm1):
        d.set_pose(T_mc, scenes.random_joints(m, 6))
        D.generate_obs_sdf(torch.from_numpy(obs).to(gpu), d, focal_length=fl[0])
    og = O.make_grid(layout[0], layout[1], layout[2], D.obs_sdf_numpy(dm))
    pred = D.raycast_prediction([dm], fl, pp, w, h)
    p_np = pred.cpu().numpy()
    da, err, norm = D.mod_to_obs_debug_maps(pred, dm, T_gc=T_mc)
    rda, rerr, rnorm = O.mod2obs_debug_maps(p_np, 0, T_mc, og)
    da0 = da.cpu().numpy().copy()
    D.mod_to_obs_debug_maps(pred, dm1, T_gc=T_mc, da=da, err=err, norm=norm)
    O.mod2obs_debug_maps(p_np, 1, T_mc, og, da=rda, err=rerr, norm=rnorm)
    assert np.array_equal(da.cpu().numpy(), rda) and np.array_equal(rda, da0)
    assert same_bits(err.cpu().numpy(), rerr) and same_bits(norm.cpu().numpy(), rnorm)
    assert not rnorm.any()                                         # the gradient map is cleared by every model


@pytest.mark.parametrize("name", ["human_hand", "barrett"])
def test_intersection_debug_error(gpu, name):
    m = scenes.articulated(name)
    om = O.OModel(m)
    dm = D.DeviceModel(m, model_id=0)
    pot = potential_matrix(name, len(m.sdfs))
    cloud = m.collision_cloud(density=1e5, seed=3)
    dm.set_pose(se3.identity(), closed_fist(m))
    _, T_mf, T_fm, _ = dm.get_transforms()
    # a second model (a sphere) pushed into the hand
    ball = scenes.sphere()
    ob = O.OModel(ball, 1)
    db = D.DeviceModel(ball, model_id=1)
    db.set_pose(se3.identity(), np.zeros(0, dtype=F))
    _, Tb_mf, Tb_fm, _ = db.get_transforms()
    centre = cloud[:, :3].mean(axis=0)
    T_ds = se3.exp(np.array([-centre[0], -centre[1], -centre[2], 0, 0, 0], dtype=F))      # destination model <- source model
    sites = torch.from_numpy(cloud).to(gpu)
    pot_d = torch.from_numpy(pot.reshape(-1)).to(gpu)
    got = D.debug_intersection_error(sites, dm, pot_d, pairs=[(T_ds, db)]).cpu().numpy()
    ref = O.debug_intersection_error(cloud, om, T_fm, T_mf, pot.reshape(-1), pairs=[(T_ds, ob, Tb_fm)])
    assert same_bits(got, ref)
    n_self = D.count_self_intersections(sites, dm, pot_d)
    n_pair = D.count_intersections(sites, dm, db, T_ds)
    assert n_self > 20 and n_pair > 20
    assert int((~np.isnan(ref)).sum()) == n_self                   # pairwise-only sites stay NaN (NaN += x), intersection.cu:433-435
    self_only = D.debug_intersection_error(sites, dm, pot_d).cpu().numpy()
    assert (ref[~np.isnan(ref)] >= self_only[~np.isnan(ref)]).all() and (ref[~np.isnan(ref)] > self_only[~np.isnan(ref)]).any()
    # empty cloud: nothing to do
    assert D.debug_intersection_error(sites[:0], dm, pot_d).numel() == 0
