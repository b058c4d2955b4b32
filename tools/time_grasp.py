import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D
D.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = configs.contact_grasp_scene(B)
tr = configs.make_grasp_tracker(sc, max_batch=B)
T, j = configs.grasp_batch_arrays(sc)
Td = torch.from_numpy(T).cuda(); jd = torch.from_numpy(j).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = tr.optimizePosesBatch(Td, jd, device_arrays=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("B=%d 40 iterations: %.2f ms -> %.1f grasps/s, %.1f GN it/s; energy mean %.4f finite %s" % (B, dt * 1e3, B / dt, B * 40 / dt, float(out[2].mean()), bool(torch.isfinite(out[2]).all())))
