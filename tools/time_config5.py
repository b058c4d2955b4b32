"""Times BASELINE config 5 (Barrett 8->4 + object, 1280x720, multi-model tracking with observations): iterations/s of
Tracker.optimizePoses."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dart_amd import configs, device as D
from test_gpu_multi_model import setup
D.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc, tr, vert, norm = setup(1280, 720, iters)
for name, fn in (("host-driven loop", tr.optimizePoses), ("device-resident loop", tr.optimizePosesDevice)):
  for rep in range(3):
    for m, (T, a) in enumerate(sc.init):
      tr.getPose(m).setTransformCameraToModel(T)
      tr.getPose(m).getReducedArticulation()[:] = a
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(vert, norm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("config 5, %s, %d iterations: %.2f ms -> %.1f GN it/s (%.3f ms/iteration)" % (name, iters, dt * 1e3, iters / dt, dt * 1e3 / iters))
