"""Host enqueue time vs GPU completion time of one dartk_optimize_pose call (C2)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D
from dart_amd.optimizer import Optimizer, synthetic_observation
D.init(0)
sc = configs.config2()
dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
dm.set_pose(sc.true_T_mc, sc.true_joints)
_, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
sc.opts.numIterations = 50
for rep in range(4):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.optimizePose(dm, vert, norm, sc.opts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("50 iterations: enqueue %.2f ms, complete %.2f ms -> %.1f us/iteration" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / 50))
