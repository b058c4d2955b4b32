"""Diagnostic (libdartk_stamps.so, tools/build_variant.py stamps -DDARTK_STAMPS): where the single-wave solve kernel of config 2
spends its time. iterInfo[5..7] = 10 ns ticks of assemble / LDL^T / rest (substitutions, pose update, forward kinematics)."""
import os, sys
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
acc = []
for _ in range(30):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts, debug=True)
    acc.append(opt.iter_info.cpu().numpy()[:, 5:8].copy())
a = np.stack(acc[5:]).reshape(-1, 3) * 0.01
print("solve kernel phases, us (median over %d launches): assemble %.2f, LDL^T + forward substitution %.2f, transpose + back substitution %.2f"
      % (len(a), np.median(a[:, 0]), np.median(a[:, 2]), np.median(a[:, 1] - a[:, 2])))
print("(the kernel's whole duration also holds its prologue loads before the first stamp and the pose update + forward kinematics after the last)")
