"""Diagnostic (library built with DARTK_EXTRA_FLAGS=-DDARTK_STAMPS): share of workgroup cycles per phase of noobs_terms_kernel."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D, _abi
D.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = configs.contact_grasp_scene(B)
tr = configs.make_grasp_tracker(sc, max_batch=B)
T, j = configs.grasp_batch_arrays(sc)
Td = torch.from_numpy(T).cuda(); jd = torch.from_numpy(j).cuda()
lib = _abi.lib()
fn = lib.dartk_debug_noobs_stamps
buf = (ctypes.c_ulonglong * 24)()
tr.optimizePosesBatch(Td.clone(), jd.clone(), device_arrays=True)
fn(buf, 1)
tr.optimizePosesBatch(Td.clone(), jd.clone(), device_arrays=True)
fn(buf, 1)
a = np.array(list(buf), dtype=np.float64).reshape(3, 8)
names = ["prologue", "acc.init", "test", "park", "rows", "accumulate", "store", "-"]
for t, tn in enumerate(["SELF", "INTER", "PRIORS"]):
    tot = a[t].sum()
    print("%-7s total %.3e cycles: " % (tn, tot) + ", ".join("%s %.1f%%" % (names[i], 100 * a[t, i] / tot) for i in range(7)))
