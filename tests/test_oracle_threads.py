"""The oracle's threaded loops (orc_set_threads > 1: bench.py's all-cores cpu_baseline leg) make the same DECISIONS as the serial
restatement every parity test uses: association lists (single and multi model) identical, in pixel order."""
import numpy as np

import oracle_lib as O
import scenes
from dart_amd import se3

F = np.float32


def test_threaded_association_equals_serial():
    m = scenes.articulated("barrett")
    om = O.OModel(m, 0)
    sph = scenes.sphere()
    os_ = O.OModel(sph, 1)
    joints = scenes.random_joints(m, 3)
    T_mc = scenes.random_T(7, base=(0, 0, -0.5))
    T_mf, T_fm = om.fk(joints)
    s_mf, s_fm = os_.fk(np.zeros(0, dtype=F))
    vert = scenes.points_near_model(m, T_mf, se3.invert(T_mc), 96, 64, 5)
    norm = scenes.random_normal_map(96, 64, 6)
    T_mc2 = se3.mul(se3.translation(0.02, 0.0, 0.01), T_mc)
    try:
        ref_s, _ = O.assoc_single(vert, norm, om, T_mc, T_fm, norm_thr=-2.0)
        ref_m, ref_c, ref_d = O.assoc_multi(vert, [om, os_], [T_mc, T_mc2], [T_fm, s_fm], [0.03, 0.03], [-0.03, -0.03], [[0, 0, 0]] * 2)
        O.set_threads(4)
        got_s, _ = O.assoc_single(vert, norm, om, T_mc, T_fm, norm_thr=-2.0)
        got_m, got_c, got_d = O.assoc_multi(vert, [om, os_], [T_mc, T_mc2], [T_fm, s_fm], [0.03, 0.03], [-0.03, -0.03], [[0, 0, 0]] * 2)
    finally:
        O.set_threads(1)
    assert len(ref_s) > 500 and ref_s.tobytes() == got_s.tobytes()
    assert np.array_equal(ref_c, got_c) and ref_c.sum() > 500
    for a, b in zip(ref_m, got_m):
        assert a.tobytes() == b.tobytes()
    assert np.array_equal(ref_d["da"], got_d["da"])
