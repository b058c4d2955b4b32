"""Pins the oracle's obs->mod normal equations (orc_ne_obs2mod: packing, residual sign, Huber branch) against an independent
fp64 construction (tests/independent_ne.py) that shares only the FD-pinned per-point Jacobian with it. The reference has no
test of its own for J^T J / J^T e; this is the strongest pin available without running the reference. CPU only."""
import numpy as np
import pytest

import independent_ne as I
import oracle_lib as O
import scenes
from dart_amd import se3

F = np.float32


def sphere_case(vert):
    m = scenes.sphere()
    om = O.OModel(m)
    T_mf, T_fm = om.fk(np.zeros(0, dtype=F))
    return m, om, se3.identity(), T_mf, T_fm, vert


def hand_case(name, w, h, seed):
    m = scenes.articulated(name)
    om = O.OModel(m)
    joints = scenes.random_joints(m, 10 + seed)
    T_mc = scenes.random_T(20 + seed, base=(0, 0, -0.5))
    T_mf, T_fm = om.fk(joints, "dh")
    vert = scenes.points_near_model(m, T_mf, se3.invert(T_mc), w, h, seed)
    return m, om, T_mc, T_mf, T_fm, vert


CASES = {
    "sphere_kat1_draw": lambda: sphere_case(scenes.kat1_vertex_map("as_compiled")),
    "sphere_64x48": lambda: sphere_case(scenes.random_vertex_map(64, 48, 1)),
    "left_hand": lambda: hand_case("left_hand", 64, 48, 0),
    "allegro": lambda: hand_case("allegro", 64, 48, 1),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_normal_equations_match_independent_construction(case):
    m, om, T_mc, T_mf, T_fm, vert = CASES[case]()
    norm = np.zeros_like(vert)
    pts, _ = O.assoc_single(vert, norm, om, T_mc, T_fm, dist_thr=0.05)
    assert len(pts) > 10
    ref, errs = I.obs2mod_system(m, om, vert, T_mc, T_fm, T_mf, pts, huber_delta=0.02)
    # the association's own error is the independently interpolated distance
    assert np.abs(errs - pts["error"]).max() < 2e-6
    n_out = int((np.abs(pts["error"]) >= 0.02).sum())
    assert n_out > 0 and n_out < len(pts), "both Huber branches must be exercised (%d of %d outliers)" % (n_out, len(pts))
    got = O.ne_obs2mod(vert, om, T_mc, T_fm, T_mf, pts, huber_delta=0.02)
    err = I.rel_err_blocks(got, ref, om.dims)
    print("%s: %d points (%d Huber outliers), oracle vs independent fp64: %.2e" % (case, len(pts), n_out, err))
    assert err < 1e-4
    # inliers only (pure squared loss) and outliers only
    for sel, what in ((np.abs(pts["error"]) < 0.02, "inliers"), (np.abs(pts["error"]) >= 0.02, "outliers")):
        sub = pts[sel]
        if len(sub) <= 10:
            continue
        ref_s, _ = I.obs2mod_system(m, om, vert, T_mc, T_fm, T_mf, sub, huber_delta=0.02)
        got_s = O.ne_obs2mod(vert, om, T_mc, T_fm, T_mf, sub, huber_delta=0.02)
        assert I.rel_err_blocks(got_s, ref_s, om.dims) < 1e-4, what


def test_sphere_system_against_analytic_geometry():
    """On the sphere the residual and gradient are known in closed form (||p|| - r, p/||p||): the SDF-based system must agree
    with the analytic one to the voxelisation error (res/8 on a 2-5 cm residual: a few per cent), which pins signs and layout
    to geometry rather than to any restatement."""
    m, om, T_mc, T_mf, T_fm, vert = sphere_case(scenes.random_vertex_map(64, 48, 1))
    pts, _ = O.assoc_single(vert, np.zeros_like(vert), om, T_mc, T_fm, dist_thr=0.05)
    p = vert.reshape(-1, 4)[pts["index"], :3].astype(np.float64)
    r = np.linalg.norm(p, axis=1)
    dist = r - 0.1
    g = p / r[:, None]
    d = 6
    JTe, JTJ, e = np.zeros(d), np.zeros((d, d)), 0.0
    for k in range(len(p)):
        gx, gy, gz = g[k]
        px, py, pz = p[k]
        J = np.array([-gx, -gy, -gz, gy * pz - gz * py, -gx * pz + gz * px, gx * py - gy * px])     # kernel_common.h:20-26
        if abs(dist[k]) < 0.02:
            JTe += J * -dist[k]; JTJ += np.outer(J, J); e += 0.5 * dist[k] ** 2
        else:
            v = -0.02 if dist[k] < 0 else 0.02
            JTe += v * -J; JTJ += np.outer(J, J); e += 0.02 * (abs(dist[k]) - 0.01)
    ana = I.pack(JTe, JTJ, e)
    got = O.ne_obs2mod(vert, om, T_mc, T_fm, T_mf, pts, huber_delta=0.02)
    # translation block of J^T e and J^T J, and e (rotation rows vanish analytically for a centred sphere: J_rot = g . (e_k x p) = 0)
    assert np.linalg.norm(got[:3] - ana[:3]) / np.linalg.norm(ana[:3]) < 0.05
    tri = [6 + i * (i + 1) // 2 + j for i in range(3) for j in range(i + 1)]
    assert np.linalg.norm(got[tri] - ana[tri]) / np.linalg.norm(ana[tri]) < 0.05
    assert abs(got[-1] - ana[-1]) / ana[-1] < 0.05
