"""GPU parity: self- and inter-model intersection terms vs the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import scenes
from dart_amd import device as D
from dart_amd import se3
from dart_amd.model import ASSET_DIR, HostModel, GEOM_SPHERE
from test_gpu_obs2mod import rel_err_blocks

F = np.float32
pytestmark = pytest.mark.gpu


def potential_matrix(name, n):
    spec = json.load(open(os.path.join(ASSET_DIR, name + ".json")))
    pot = np.zeros((n, n), dtype=np.int32)
    mat = np.asarray(spec["intersection_potential"], dtype=np.int32)
    k = min(n, mat.shape[0])
    pot[:k, :k] = mat[:k, :k]      # allegro ships a 13x13 file for 17 SDFs (Appendix A-9): the rest stays 0 here
    return pot


FLEX = {"human_hand": 1.0, "allegro": 0.9, "barrett": 0.75}     # fraction of each joint's range that makes links collide


def closed_fist(m):
    lim = np.asarray(m.joint_limits, dtype=F)
    return (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * F(FLEX[m.name])).astype(F)


@pytest.mark.parametrize("name", ["human_hand", "allegro", "barrett"])
def test_self_intersection(gpu, name):
    m = scenes.articulated(name)
    om = O.OModel(m)
    dm = D.DeviceModel(m, model_id=0)
    pot = potential_matrix(name, len(m.sdfs))
    cloud = m.collision_cloud(density=1e5, seed=3)
    assert len(cloud) > 1000
    joints = closed_fist(m)
    dm.set_pose(se3.identity(), joints)
    _, T_mf, T_fm, _ = dm.get_transforms()
    sites = torch.from_numpy(cloud).to(gpu)
    pot_d = torch.from_numpy(pot.reshape(-1)).to(gpu)
    res = D.norm_eqns_self_intersection(sites, dm, pot_d).cpu().numpy()
    ref, n = O.ne_self_intersection(cloud, om, T_fm, T_mf, pot.reshape(-1))
    assert n > 20, "scene must actually collide (%d)" % n
    assert D.count_self_intersections(sites, dm, pot_d) == n == O.count_self_intersections(cloud, om, T_fm, T_mf, pot.reshape(-1))
    assert rel_err_blocks(res, ref, m.dims - 6) < 1e-4
    # no potential pairs => nothing
    zero = torch.zeros_like(pot_d)
    assert not D.norm_eqns_self_intersection(sites, dm, zero).cpu().numpy().any()
    assert D.count_self_intersections(sites, dm, zero) == 0
    # empty cloud
    assert not D.norm_eqns_self_intersection(sites[:0], dm, pot_d).cpu().numpy().any()


def box_like_object():
    m = HostModel(version=1, name="object")
    m._geom_specs = [{"frame": 0, "type": GEOM_SPHERE, "scale": [0.05, 0.04, 0.06], "trans": [0, 0, 0], "rot": [0, 0, 0]}]
    m.add_geometry(0, GEOM_SPHERE, [0.05, 0.04, 0.06], [0, 0, 0], [0, 0, 0])
    return m.voxelize(0.005, 0.10)


def test_inter_model_intersection(gpu):
    hand = scenes.articulated("human_hand")
    obj = box_like_object()
    oh, oo = O.OModel(hand, 0), O.OModel(obj, 1)
    dh, do = D.DeviceModel(hand, model_id=0), D.DeviceModel(obj, model_id=1)
    joints = scenes.random_joints(hand, 5)
    T_mc_h = se3.invert(se3.translation(0.0, 0.0, 0.5))
    T_mc_o = se3.invert(se3.mul(se3.translation(-0.08, 0.01, 0.5), se3.exp([0, 0, 0, 0.2, -0.1, 0.3])))
    dh.set_pose(T_mc_h, joints)
    do.set_pose(T_mc_o, np.zeros(0, dtype=F))
    _, h_mf, h_fm, _ = dh.get_transforms()
    _, o_mf, o_fm, _ = do.get_transforms()
    hand_cloud = hand.collision_cloud(1e5, seed=1)
    obj_cloud = obj.collision_cloud(1e5, seed=2)
    for (src, dst, osrc, odst, cloud, s_mc, d_mc, s_fm, s_mf, d_fm, d_mf) in [
            (dh, do, oh, oo, hand_cloud, T_mc_h, T_mc_o, h_fm, h_mf, o_fm, o_mf),
            (do, dh, oo, oh, obj_cloud, T_mc_o, T_mc_h, o_fm, o_mf, h_fm, h_mf)]:
        T_ds = se3.mul(d_mc, se3.invert(s_mc))        # optimizer.cpp:396-397
        T_sd = se3.invert(T_ds)
        sites = torch.from_numpy(cloud).to(gpu)
        res = D.norm_eqns_intersection(sites, src, dst, T_ds, T_sd).cpu().numpy()
        ref, n = O.ne_intersection(cloud, T_ds, T_sd, osrc, s_fm, s_mf, odst, d_fm, d_mf)
        assert n > 20, "scene must actually collide (%d)" % n
        assert D.count_intersections(sites, src, dst, T_ds) == n == O.count_intersections(cloud, T_ds, osrc, s_mf, odst, d_fm)
        assert rel_err_blocks(res, ref, osrc.dims) < 1e-4


def test_reduced_intersection_variants(gpu):
    m = scenes.articulated("barrett")
    om = O.OModel(m)
    dm = D.DeviceModel(m, model_id=0)
    pot = potential_matrix("barrett", len(m.sdfs))
    cloud = m.collision_cloud(1e5, seed=3)
    dm.set_pose(se3.identity(), closed_fist(m))
    _, T_mf, T_fm, _ = dm.get_transforms()
    # the Barrett coupling pattern of grasp_analyzer.cpp:409-417 (all ones there) with Barrett.xml:17's 1/3 gear ratios as values
    A = np.zeros((8, 4), dtype=F)
    for f_, r, c in [(0, 0, 1), (3, 0, 1), (1, 1, 1), (2, 1, 1 / 3), (4, 2, 1), (5, 2, 1 / 3), (6, 3, 1), (7, 3, 1 / 3)]:
        A[f_, r] = c
    mapping = np.array([0, 1, 1, 0, 2, 2, 3, 3], dtype=np.int32)
    sites = torch.from_numpy(cloud).to(gpu)
    pot_d = torch.from_numpy(pot.reshape(-1)).to(gpu)
    for red_d, red_o in [(("linear", torch.from_numpy(A.reshape(-1)).to(gpu), 10), ("linear", A.reshape(-1), 10)),
                         (("map", torch.from_numpy(mapping).to(gpu), 10), ("map", mapping, 10))]:
        res = D.norm_eqns_self_intersection(sites, dm, pot_d, reduction=red_d).cpu().numpy()
        ref, n = O.ne_self_intersection(cloud, om, T_fm, T_mf, pot.reshape(-1), reduction=red_o)
        assert n > 5
        assert rel_err_blocks(res, ref, 4) < 1e-4
    # inter-model reduced: barrett cloud against an object
    obj = box_like_object()
    oo, do = O.OModel(obj, 1), D.DeviceModel(obj, model_id=1)
    T_mc_o = se3.invert(se3.translation(0.0, 0.02, 0.09))
    do.set_pose(T_mc_o, np.zeros(0, dtype=F))
    _, o_mf, o_fm, _ = do.get_transforms()
    T_ds = T_mc_o.copy()
    T_sd = se3.invert(T_ds)
    for red_d, red_o in [(("linear", torch.from_numpy(A.reshape(-1)).to(gpu), 10), ("linear", A.reshape(-1), 10)),
                         (("map", torch.from_numpy(mapping).to(gpu), 10), ("map", mapping, 10))]:
        res = D.norm_eqns_intersection(sites, dm, do, T_ds, T_sd, reduction=red_d).cpu().numpy()
        ref, n = O.ne_intersection(cloud, T_ds, T_sd, om, T_fm, T_mf, oo, o_fm, o_mf, reduction=red_o)
        assert n > 5
        assert rel_err_blocks(res, ref, 10) < 1e-4
