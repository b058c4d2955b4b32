"""bench.py's single-frame loop of the multi-model leg (the frames of configs.config5_batch one call at a time), frame by frame."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D
D.init(0)
msc = configs.config5(1280, 720, 0)
mtr = configs.make_multi_tracker(msc)
mtr.getOptions().numIterations = 10
f_idx = list(range(16))
mvert, mnorm, mT0, ma0 = configs.config5_batch(mtr, msc, f_idx)
def frame(k):
    for m in range(2):
        mtr.getPose(m).setTransformCameraToModel(mT0[k, m])
    mtr.getPose(0).getReducedArticulation()[:] = ma0[k]
    mtr.optimizePosesDevice(mvert[k], mnorm[k], readback=False)
frame(0); torch.cuda.synchronize()
for rep in range(2):
    ts = []
    for k in range(8):
        t0 = time.perf_counter(); frame(k); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print("synchronised after every frame, us:", " ".join("%.0f" % t for t in ts))
for n in (8, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        frame(k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d frames back to back: %.0f us per frame, %.0f GN it/s" % (n, dt / n * 1e6, n * 10 / dt))
print("valid pixels per frame:", [int((mvert[k, :, :, 3] > 0).sum().item()) for k in range(4)])
