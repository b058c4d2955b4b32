"""Diagnostic (libdartk_stamps.so, tools/build_variant.py stamps -DDARTK_STAMPS): where the workgroups of config 2's mod->obs
launch (replay variant) spend their time: 10 ns realtime ticks at the phase boundaries of thread 0's wave (no waits added: a
stamp is reached when the instructions before it have been issued, so a load's latency shows up where its value is first needed)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ".")
from dart_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libdartk_stamps.so")
import torch
from dart_amd import configs, device as D
from dart_amd.optimizer import Optimizer, synthetic_observation

D.init(0)
sc = configs.config2()
dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
dm.set_pose(sc.true_T_mc, sc.true_joints)
depth, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
buf = (ctypes.c_longlong * (1024 * 8))()
fn = _abi.lib().dartk_debug_m2o_phases
names = ["entry", "prologue issued", "prediction pixel in hand", "replay done", "prediction stored (issued)", "value + label",
         "rows built (barrier)", "accumulated + slab stored (issued)"]
n_wg = (sc.pred_width * sc.pred_height + 127) // 128
for rep in range(4):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts)
    fn(buf)
    st = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8)[:n_wg].astype(np.float64) * 0.01
    t0 = st[:, 0].min()
    rel = st - t0
    print("run %d: %d workgroups, first entry -> last exit %.2f us; entries spread over %.2f us" % (rep, n_wg, rel[:, 7].max(), rel[:, 0].max()))
    for i in range(1, 8):
        ph = st[:, i] - st[:, i - 1]
        print("  %-36s phase: median %5.2f  p90 %5.2f  max %5.2f   |  reached at: median %5.2f  max %5.2f" %
              (names[i], np.median(ph), np.percentile(ph, 90), ph.max(), np.median(rel[:, i]), rel[:, i].max()))
    last = np.argsort(-rel[:, 7])[:6]
    print("  last workgroups out:", [(int(i), [round(float(v), 1) for v in rel[i]]) for i in last])
