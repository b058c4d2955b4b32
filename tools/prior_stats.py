"""Diagnostic (python tools/build_variant.py pstats -DDARTK_COUNT_LOOKUPS -DDARTK_PRIOR_STATS): per-wave statistics of the contact-prior
nearest-SDF loop: wave-union iterations, lane-level sphere passes, per-wave maximum of a lane's passes, value lookups."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import _abi
_abi.LIB_PATH = os.path.join(R, "dart_amd", "libdartk_pstats.so")
from dart_amd import configs, device as D
D.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = configs.contact_grasp_scene(B)
tr = configs.make_grasp_tracker(sc, max_batch=B)
tr.getOptions().numIterations = 10
T, j = configs.grasp_batch_arrays(sc)
Td = torch.from_numpy(T).cuda(); jd = torch.from_numpy(j).cuda()
cnt = (C.c_uint64 * 8)()
L = _abi.lib()
L.dartk_lookup_counters(cnt, 1)
tr.optimizePosesBatch(Td.clone(), jd.clone(), device_arrays=True)
torch.cuda.synchronize()
cnt = (C.c_uint64 * 8)()
L.dartk_lookup_counters(cnt, 1)
c = [int(v) for v in cnt]
waves = c[4]
print("waves %d (each up to 64 priors x 16 SDFs); per wave: union iterations %.2f, max-per-lane %.2f, lane passes %.1f (dense = %.2f iterations), in-loop value lookups %.1f; all prior lookups %.1f"
      % (waves, c[1] / waves, c[3] / waves, c[2] / waves, c[2] / waves / 64, c[5] / waves, c[7] / waves))
