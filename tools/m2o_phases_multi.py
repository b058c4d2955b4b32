"""Diagnostic (libdartk_stamps.so): the phase stamps of the mod->obs workgroups (replay of every model's traces inside the term,
REPLAY == 2) in the second shared launch of the config-5 single-frame loop. The two models' segments write the same stamp slots
(workgroup index within the segment): what is read back is the segment that ran last, the object's."""
import ctypes, os, sys
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
buf = (ctypes.c_longlong * (1024 * 8))()
fn = _abi.lib().dartk_debug_m2o_phases
names = ["entry", "prologue issued", "prediction pixel in hand", "replay done", "prediction stored (issued)", "value + label",
         "rows built (barrier)", "accumulated + slab stored (issued)"]
n_wg = (tr.pred_width * tr.pred_height + 127) // 128
for rep in range(3):
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
    tr.optimizePosesDevice(vert, norm)
    fn(buf)
    st = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8)[:n_wg].astype(np.float64) * 0.01
    ok = st[:, 0] > 0
    st = st[ok]
    t0 = st[:, 0].min()
    rel = st - t0
    print("run %d: %d workgroups, first entry -> last exit %.2f us; entries spread over %.2f us" % (rep, len(st), rel[:, 7].max(), rel[:, 0].max()))
    for i in range(1, 8):
        ph = st[:, i] - st[:, i - 1]
        print("  %-36s phase: median %5.2f  p90 %5.2f  max %5.2f   |  reached at: median %5.2f  max %5.2f" %
              (names[i], np.median(ph), np.percentile(ph, 90), ph.max(), np.median(rel[:, i]), rel[:, i].max()))
    last = np.argsort(-rel[:, 7])[:4]
    print("  last workgroups out:", [(int(i), [round(float(v), 1) for v in rel[i]]) for i in last])
