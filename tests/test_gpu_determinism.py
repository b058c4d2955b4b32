"""The device paths use no float atomics and fixed-order reductions: repeating a call must give bit-identical results
(a race between workgroups, streams or LDS phases shows up here as a difference)."""
import numpy as np
import pytest
import torch

from dart_amd import configs
from dart_amd import device as D
from dart_amd.optimizer import Optimizer, synthetic_observation

pytestmark = pytest.mark.gpu


def test_optimize_pose_is_bitwise_repeatable(gpu):
    sc = configs.config2()
    dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
    dm.set_pose(sc.true_T_mc, sc.true_joints)
    _, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    sc.opts.numIterations = 4
    runs = []
    for rep in range(6):
        dm.set_pose(sc.init_T_mc, sc.init_joints)
        opt.optimizePose(dm, vert, norm, sc.opts, debug=True)
        T, _, _, j = dm.get_transforms()
        runs.append((T.copy(), j.copy(), opt.iter_info.cpu().numpy().copy(), opt.first_system.cpu().numpy().copy(), opt.prediction.cpu().numpy().copy()))
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert np.array_equal(a, b)


def test_batched_refinement_is_bitwise_repeatable(gpu):
    sc = configs.contact_grasp_scene(96, n_attract=200, n_repulse=600)
    tr = configs.make_grasp_tracker(sc, max_batch=96)
    tr.getOptions().numIterations = 6
    T, j = configs.grasp_batch_arrays(sc)
    outs = [tr.optimizePosesBatch(T, j, debug=True) for _ in range(4)]
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
    # a candidate's result does not depend on its position in the batch or on the batch size (96 -> large-batch block layout,
    # 8 -> one workgroup per tile: different partial-sum grouping, so only agreement to rounding is expected there)
    perm = np.random.default_rng(0).permutation(96)
    Tp, jp, ep, _, _ = tr.optimizePosesBatch(T[perm], j[perm])
    assert np.array_equal(Tp, outs[0][0][perm]) and np.array_equal(jp, outs[0][1][perm]) and np.array_equal(ep, outs[0][2][perm])
