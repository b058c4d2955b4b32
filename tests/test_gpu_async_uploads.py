"""GPU: entry points that upload small host blocks without waiting for the stream -- dartk_assoc_multi (argument blocks through a
ring of pinned slots) and dartk_model_set_pose (the pose in the kernel arguments of its launch) -- stay correct when many calls are
in flight: more calls than the ring has slots, each with its own arguments, none of them synchronised before the end."""
import ctypes as C

import numpy as np
import pytest
import torch

import scenes
from dart_amd import _abi, device as D, se3
from dart_amd._abi import check, lib

F = np.float32
pytestmark = pytest.mark.gpu


def test_assoc_multi_calls_in_flight_keep_their_own_arguments(gpu):
    m = scenes.articulated("left_hand")
    dm = D.DeviceModel(m, model_id=0)
    rng = np.random.default_rng(2)
    lim = np.asarray(m.joint_limits, dtype=F).reshape(-1, 2)
    joints = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * 0.3).astype(F)
    T_mc = se3.invert(se3.translation(0.0, 0.0, 0.5))
    dm.set_pose(T_mc, joints)
    _, T_mf, _, _ = dm.get_transforms()
    vert = scenes.points_near_model(m, T_mf, se3.invert(T_mc), 160, 120, 6)
    v_d = torch.from_numpy(vert).to(gpu)
    n_d = torch.zeros_like(v_d)
    n_calls = 24                                   # three times the ring
    thr = np.linspace(0.002, 0.03, n_calls).astype(F)
    # reference: one call at a time, synchronised
    ref = []
    for t in thr:
        _, counts, _ = D.error_and_data_association_multi(v_d, n_d, [dm], [t], [-1e9], [(0, 0, 0)])
        ref.append(int(counts[0]))
    assert len(set(ref)) > 10                      # the thresholds really give different answers
    # the same calls back to back, each with its own output, no synchronisation in between
    h, w = vert.shape[:2]
    lasts = [torch.zeros(1, dtype=torch.int32, device=gpu) for _ in range(n_calls)]
    pts = [D.new_pts(h * w, gpu) for _ in range(n_calls)]
    handles = (C.c_void_p * 1)(dm.handle)
    keep = []
    torch.cuda.synchronize()
    for k in range(n_calls):
        dt, po, pn = np.array([thr[k]], dtype=F), np.array([-1e9], dtype=F), np.zeros(3, dtype=F)
        table = (C.c_void_p * 1)(C.c_void_p(pts[k].data_ptr()))
        keep.append((dt, po, pn, table))
        check(lib().dartk_assoc_multi(D._ptr(v_d), D._ptr(n_d), w, h, 1, handles, _abi.fptr(dt), _abi.fptr(po), _abi.fptr(pn), table,
                                      D._ptr(lasts[k]), None, None, D._stream()))
        dt[0] = -1.0                               # the host arrays are free for reuse as soon as the call returns
    torch.cuda.synchronize()
    got = [int(l.cpu().numpy()[0]) for l in lasts]
    assert got == ref


def test_set_pose_back_to_back(gpu):
    """Twelve poses set one after the other without a wait: every forward-kinematics launch carries its own pose, and the last
    one wins; a pose read back between two sets is the one that was set before the read."""
    m = scenes.articulated("allegro")
    dm = D.DeviceModel(m, model_id=0)
    lim = np.asarray(m.joint_limits, dtype=F).reshape(-1, 2)
    rng = np.random.default_rng(4)
    poses = []
    for k in range(12):
        j = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.random(len(lim))).astype(F)
        T = se3.mul(se3.translation(*(0.01 * rng.normal(size=3))), se3.exp(np.concatenate([np.zeros(3), 0.2 * rng.normal(size=3)]).astype(F)))
        poses.append((T, j))
    for T, j in poses:
        Tb, jb = T.copy(), j.copy()
        dm.set_pose(Tb, jb)
        Tb[:] = 7.0; jb[:] = 7.0                   # the caller's arrays may change at once
    T_mc, T_mf, T_fm, joints = dm.get_transforms()
    assert np.array_equal(T_mc, poses[-1][0]) and np.array_equal(joints, poses[-1][1])
    ref_mf = np.asarray(m.forward_kinematics(poses[-1][1], "dh"), dtype=F).reshape(-1, 12)
    assert np.abs(T_mf.reshape(-1, 12) - ref_mf).max() < 1e-5
    dm.set_pose(*poses[3])
    T_mc, _, _, joints = dm.get_transforms()
    assert np.array_equal(T_mc, poses[3][0]) and np.array_equal(joints, poses[3][1])
