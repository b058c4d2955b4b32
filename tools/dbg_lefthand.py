import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from dart_amd import configs, device as D
import test_gpu_optimize_pose as T
D.init(0)
np.set_printoptions(linewidth=200, precision=5, suppress=True)
sc = configs.config2_left_hand(320, 240)
for it in (1, 2, 3, 5):
    sc = configs.config2_left_hand(320, 240)
    dm, g_T, g_j, g_info, g_first, o_T, o_j, o_info, o_first = T.run_scene(sc, torch.device("cuda:0"), iters=it)
    print("iters", it, "dj", np.abs(g_j - o_j).max(), "argmax", np.abs(g_j - o_j).argmax())
    print(" g_info", g_info[:, :5].tolist())
    print(" o_info", o_info[:, :5].tolist())
    print(" dq", (g_j - o_j))
