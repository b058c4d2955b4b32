"""Host enqueue time against device time of one dartk_optimize_poses call (config 5, one frame): is the multi-model loop bound by
the host's launches?"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dart_amd import _abi, device as D
from test_gpu_multi_model import setup
D.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc, tr, vert, norm = setup(1280, 720, iters)
L = _abi.lib()
orig = L.dartk_optimize_poses
stamps = []
class Wrap:
    def __call__(self, *a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = orig(*a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        stamps.append((t1 - t0, t2 - t0))
        return rc
L.dartk_optimize_poses = Wrap()
for rep in range(8):
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm)
e = np.array(stamps[3:]) * 1e6
print("dartk_optimize_poses, %d iterations: host enqueue %.0f us (%.1f per iteration), enqueue + drain %.0f us (%.1f per iteration)"
      % (iters, np.median(e[:, 0]), np.median(e[:, 0]) / iters, np.median(e[:, 1]), np.median(e[:, 1]) / iters))
