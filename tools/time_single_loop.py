"""Where a frame of the config-5 single-frame loop (Tracker.optimizePosesDevice, readback=False) spends its time: the loop as
bench.py runs it (every frame starts from its own host-side start pose), the same without the per-frame pose upload (a tracker
carries the pose over on the device: from_device=True), and the host's share of a call."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dart_amd import device as D
from test_gpu_multi_model import setup
D.init(0)
iters = 10
sc, tr, vert, norm = setup(1280, 720, iters)
T0 = [np.array(T, dtype=np.float32) for T, a in sc.init]
a0 = [np.array(a, dtype=np.float32) for T, a in sc.init]
def reset():
    for m in range(2):
        tr.getPose(m).setTransformCameraToModel(T0[m])
        if len(a0[m]):
            tr.getPose(m).getReducedArticulation()[:] = a0[m]
N = 40
for mode in ("reset every frame (bench.py)", "pose carried over"):
    for rep in range(3):
        reset(); tr.optimizePosesDevice(vert, norm, readback=False)
        torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        for k in range(N):
            if mode.startswith("reset"): reset()
            h0 = time.perf_counter()
            tr.optimizePosesDevice(vert, norm, readback=False, from_device=not mode.startswith("reset"))
            host += time.perf_counter() - h0
        hostAll = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-30s %7.1f us per frame (%6.0f GN it/s); host returns after %6.1f us per frame, %6.1f us of it inside optimizePosesDevice"
          % (mode, dt / N * 1e6, N * iters / dt, hostAll / N * 1e6, host / N * 1e6))
