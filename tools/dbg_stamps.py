import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dart_amd import configs, device as D
from dart_amd.optimizer import Optimizer, synthetic_observation
D.init(0)
sc = configs.config2()
dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
dm.set_pose(sc.true_T_mc, sc.true_joints)
_, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
for _ in range(3):
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts, debug=True)
torch.cuda.synchronize()
print("stamps (us) assemble / ldlt / update+fk per iteration:")
print(opt.iter_info.cpu().numpy()[:, 5:8] / 100.0)
import ctypes
from dart_amd import _abi
buf = (ctypes.c_longlong * 9)()
fn = getattr(_abi.lib(), "dartk_debug_ray_steps", None)
if fn is not None:
    fn(buf)
    dm.set_pose(sc.init_T_mc, sc.init_joints)
    opt.optimizePose(dm, vert, norm, sc.opts)
    fn(buf)
    n_it = sc.opts.numIterations
    if buf[3]:
        print("longest wave (last launch group): %d passes; cycles per pass: coordinates %.0f, load wait %.0f, interpolation + step %.0f; "
              "all waves: %.0f cycles per pass" % (buf[3], buf[4] / buf[3], buf[5] / buf[3], buf[6] / buf[3], buf[8] / max(buf[7], 1)))
    print("ray-cast per iteration: longest march %d steps, %.0f marched rays, %.1f steps/ray" % (buf[0], buf[2] / n_it, buf[1] / max(buf[2], 1)))
