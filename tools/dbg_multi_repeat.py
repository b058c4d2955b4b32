import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from dart_amd import device as D
from test_gpu_multi_model import setup
D.init(0)
sc, tr, vert, norm = setup(640, 360, 6)
runs = []
for _ in range(5):
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, debug=True)
    runs.append(([tr.getPose(m).getTransformCameraToModel().copy() for m in range(2)], tr.getPose(0).getReducedArticulation().copy(), tr.iteration_summaries.copy()))
for T, a, info in runs[1:]:
    print("counts equal", np.array_equal(info[:, :, 3:5], runs[0][2][:, :, 3:5]), [float(np.abs(T[m] - runs[0][0][m]).max()) for m in range(2)], float(np.abs(a - runs[0][1]).max()))
    if not np.array_equal(info[:, :, 3:5], runs[0][2][:, :, 3:5]):
        print(info[:, :, 3:5].reshape(-1), runs[0][2][:, :, 3:5].reshape(-1))
