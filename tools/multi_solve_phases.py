"""Diagnostic (libdartk_stamps.so, tools/build_variant.py stamps -DDARTK_STAMPS): where the single-wave solve kernel of the config-5
single-frame loop spends its time. iterInfo[it][0][5..7] = 10 ns ticks of assembly / priors + back substitution / pose updates + FK."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from dart_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libdartk_stamps.so")
import torch
from dart_amd import device as D
from test_gpu_multi_model import setup
D.init(0)
sc, tr, vert, norm = setup(1280, 720, 10)
acc = []
for rep in range(12):
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm, debug=True)
    acc.append(tr.iteration_summaries[:, 0, 5:8].copy())
a = np.stack(acc[2:]).reshape(-1, 3) * 0.01
print("multi_solve_kernel phases, us (median over %d launches): assembly %.2f, priors + back substitution %.2f, pose updates + FK %.2f"
      % (len(a), np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2])))
