"""Times dartk_optimize_poses_batch on config 5 (profiling aid): python tools/time_batch.py [frames] [steps] [iters]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dart_amd import configs, device as D

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D.init(0)
dev = torch.device("cuda:0")
sc = configs.config5(1280, 720, 0)
tr = configs.make_multi_tracker(sc)
tr.getOptions().numIterations = iters
vert, norm, T0, a0 = configs.config5_batch(tr, sc, range(frames))
T0_d, a0_d = torch.from_numpy(T0).to(dev), torch.from_numpy(a0).to(dev)
tr.optimizePosesBatch(vert, norm, T0_d, a0_d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.optimizePosesBatch(vert, norm, T0_d, a0_d)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d frames x %d iterations: %.2f ms per call, %.1f us per frame-iteration, %.0f GN it/s" %
      (frames, iters, 1000 * dt / steps, 1e6 * dt / steps / frames / iters, frames * iters * steps / dt))
