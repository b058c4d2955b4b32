"""Per-call set-up against per-iteration cost of the single-frame loops: the time of a call with 1 .. 20 Gauss-Newton iterations,
calls enqueued back to back (config 2 through Optimizer.optimizePose as bench.py's headline runs it; config 5 through
Tracker.optimizePosesDevice as its single-frame loop does). The slope is an iteration, the intercept the call's own set-up."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D
from dart_amd.optimizer import Optimizer, synthetic_observation
D.init(0)

def fit(name, call, set_iters, counts=(1, 2, 5, 10, 20), n=32):
    pts = []
    for it in counts:
        set_iters(it)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                call(k)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
        pts.append((it, dt))
    (i1, t1), (i2, t2) = pts[-2], pts[-1]
    slope = (t2 - t1) / (i2 - i1)
    print("%s: %s -> %.1f us per iteration, %.0f us per call besides" % (name, ", ".join("%d it %.0f us" % p for p in pts), slope, t2 - slope * i2))

sc = configs.config2()
dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
dm.set_pose(sc.true_T_mc, sc.true_joints)
depth, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
def c2(k):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts)
def c2_iters(it):
    sc.opts.numIterations = it
fit("config 2 (headline)", c2, c2_iters)

msc = configs.config5(1280, 720, 0)
mtr = configs.make_multi_tracker(msc)
mvert, mnorm, mT0, ma0 = configs.config5_batch(mtr, msc, list(range(8)))
def c5(k):
    k %= 8
    for m in range(2):
        mtr.getPose(m).setTransformCameraToModel(mT0[k, m])
    mtr.getPose(0).getReducedArticulation()[:] = ma0[k]
    mtr.optimizePosesDevice(mvert[k], mnorm[k], readback=False)
def c5_iters(it):
    mtr.getOptions().numIterations = it
fit("config 5 (single-frame loop)", c5, c5_iters)
