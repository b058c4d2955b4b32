"""Diagnostic (libdartk_stamps.so): rays marched and march steps of the batch's trace launches (config 5, 64 frames, 1 iteration)."""
import ctypes, os, sys
sys.path.insert(0, ".")
from dart_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libdartk_stamps.so")
import numpy as np, torch
from dart_amd import configs, device as D
D.init(0)
dev = torch.device("cuda:0")
sc = configs.config5(1280, 720, 0)
tr = configs.make_multi_tracker(sc)
tr.getOptions().numIterations = 1
vert, norm, T0, a0 = configs.config5_batch(tr, sc, range(64))
buf = (ctypes.c_longlong * 9)()
fn = _abi.lib().dartk_debug_ray_steps
fn(buf)
tr.optimizePosesBatch(vert, norm, torch.from_numpy(T0).to(dev), torch.from_numpy(a0).to(dev))
fn(buf)
print("64 frames, 1 iteration, both models: longest march %d steps, %d rays marched, %d steps in all (%.1f per ray)" % (buf[0], buf[2], buf[1], buf[1] / max(buf[2], 1)))
