"""How long does the host take to ENQUEUE one dartk_optimize_pose call (config 2, 20 iterations) compared with the device time
of the call? If enqueueing takes as long as the device work, the loop is launch-bound and a captured graph pays."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from dart_amd import configs, device as D
from dart_amd.optimizer import Optimizer, synthetic_observation

D.init(0)
sc = configs.config2()
dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
dm.set_pose(sc.true_T_mc, sc.true_joints)
depth, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
for _ in range(20):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(50):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.optimizePose(dm, vert, norm, sc.opts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print("one call, %d iterations: enqueue median %.1f us, enqueue + drain median %.1f us" % (sc.opts.numIterations, 1e6 * np.median(enq), 1e6 * np.median(tot)))
# back to back (what bench.py times): the host runs ahead of the device only if enqueueing is the shorter of the two
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("100 calls back to back: host done after %.1f us per call, device done after %.1f us per call" % (1e4 * (t1 - t0), 1e4 * (t2 - t0)))
