import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from dart_amd import configs, device as D
import test_gpu_contact_grasp as T
D.init(0)
np.set_printoptions(linewidth=220, precision=6, suppress=True)
sc = T.scene(4)
tr = configs.make_grasp_tracker(sc)
Tm, j = configs.grasp_batch_arrays(sc)
tr.getOptions().numIterations = 8
gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(Tm[:1], j[:1], debug=True)
oT, oj, oe, oinfo, ofirst = T.oracle_run(sc, tr, Tm[0, 0], j[0], 8)
print("gpu  info", ginfo[0, :, :2].T)
print("orc  info", oinfo[:, :2].T)
for it in range(1, 5):
    tr.getOptions().numIterations = it
    gT, gj, ge, _, _ = tr.optimizePosesBatch(Tm[:1], j[:1], debug=True)
    oT, oj, oe, _, _ = T.oracle_run(sc, tr, Tm[0, 0], j[0], it)
    print(it, "dT hand", np.abs(gT[0, 0] - oT[0]).max(), "dT obj", np.abs(gT[0, 1] - oT[1]).max(), "dq", np.abs(gj[0] - oj[0]).max(), ge[0], oe)
