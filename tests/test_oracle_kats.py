"""Pins the CPU oracle against the reference's own known-answer tests, restated here:
  KAT 1  test/test_obs_to_mod_kernels.cpp:14-109  sphere association: error ~ |p|-r within res/8, DA in {0,-1}
  KAT 2  test/test_voxelize.cpp:11-48             sphere SDF: interpolation at voxel centres is exact; value*res ~ |c|-r
  KAT 3  test/test_model_jacobian.{cpp,cu}        analytic Jacobians vs central finite differences (delta 1e-3, tol 1e-4)
  KAT 4  test/test_pose_reduction.cpp:8-46        NullReduction aliasing / limits (host logic)
These run on the CPU (no GPU)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import scenes
from dart_amd import se3
from dart_amd.pose import NullReduction, Pose

F = np.float32


def test_kat2_sphere_voxelize_exact_at_centres():
    m = scenes.sphere()
    g = m.sdfs[0]
    assert g.resolution == F(0.005)
    assert g.dim == (80, 80, 80)
    og = O.make_grid(g.dim, g.offset, g.resolution, g.data)
    rng = np.random.default_rng(0)
    # every voxel with x,y,z < dim-1, like the reference loop; sampled densely (all of one octant + random rest)
    idx = np.stack(np.meshgrid(np.arange(0, 79, 3), np.arange(0, 79, 3), np.arange(0, 79, 3), indexing="ij"), -1).reshape(-1, 3)
    idx = np.concatenate([idx, rng.integers(0, 79, size=(20000, 3))])
    for x, y, z in idx:
        c = np.array([x + 0.5, y + 0.5, z + 0.5], dtype=F)
        val = g.data[z, y, x]
        assert O.lib().orc_grid_value_interp(C.byref(og), O.f(c)) == val
        cw = g.resolution * c + g.offset
        assert abs(val * F(0.005) - (np.linalg.norm(cw.astype(np.float64)) - 0.1)) < 1e-6


def test_kat1_sphere_association_matches_analytic():
    m = scenes.sphere()
    om = O.OModel(m)
    T_mf, T_fm = om.fk(np.zeros(0, dtype=F))
    for seed, (w, h) in enumerate([(10, 10), (64, 48)]):
        vert = scenes.random_vertex_map(w, h, seed)
        norm = np.zeros_like(vert)          # the reference test leaves the normal map at zero (w == 0: no normal test)
        pts, dbg = O.assoc_single(vert, norm, om, se3.identity(), T_fm, dist_thr=0.05)
        n_assoc = 0
        for i in range(w * h):
            p = vert.reshape(-1, 4)[i]
            da, err = dbg["da"].reshape(-1)[i], dbg["err"].reshape(-1)[i]
            if p[3] == 0:
                assert da == -1 and np.isnan(err)
                continue
            dist = np.linalg.norm(p[:3].astype(np.float64)) - 0.1
            if abs(dist) > 0.05:
                # the SDF box (padding 0.1) ends 0.2 m from the centre; beyond the threshold nothing is associated
                if abs(abs(dist) - 0.05) > 0.005 / 8:
                    assert da == -1 and np.isnan(err)
            else:
                if abs(abs(dist) - 0.05) > 0.005 / 8:
                    assert da == 0
                    assert abs(dist - err) < 0.005 / 8
                    n_assoc += 1
        assert n_assoc > 0
        assert len(pts) == (dbg["da"] >= 0).sum()
        assert np.all(np.diff(pts["index"]) > 0)


def kat1_reference_asserts(vert, da, err, dist_thr=0.05, radius=0.1, res=0.005):
    """The checks of test_obs_to_mod_kernels.cpp:87-102, verbatim in meaning: invalid pixel or |dist| > threshold => DA -1 and
    NaN error; otherwise DA 0 and |dist - error| <= res/8 (EXPECT_NEAR). dist = length(float3) - radius in fp32 like the test."""
    n_assoc = 0
    p = vert.reshape(-1, 4)
    da, err = np.asarray(da).reshape(-1), np.asarray(err).reshape(-1)
    for i in range(len(p)):
        if p[i, 3] == 0:
            assert da[i] == -1 and np.isnan(err[i]), i
            continue
        x, y, z = p[i, :3]
        dist = F(np.sqrt(F(F(F(x * x) + F(y * y)) + F(z * z)))) - F(radius)
        if abs(dist) > F(dist_thr):
            assert da[i] == -1 and np.isnan(err[i]), (i, dist, da[i], err[i])
        else:
            assert da[i] == 0, (i, dist, da[i])
            assert abs(float(dist) - float(err[i])) <= res / 8, (i, dist, err[i])
            n_assoc += 1
    return n_assoc


@pytest.mark.parametrize("order", ["as_compiled", "left_to_right"])
def test_kat1_reference_draw(order):
    """KAT 1 on the reference's own inputs (libstdc++ draw, tests/golden/kat1_inputs.json) with the reference's asserts."""
    m = scenes.sphere()
    om = O.OModel(m)
    _, T_fm = om.fk(np.zeros(0, dtype=F))
    vert = scenes.kat1_vertex_map(order)
    assert vert.shape == (10, 10, 4) and int((vert[..., 3] > 0).sum()) == 80
    norm = np.zeros_like(vert)
    pts, dbg = O.assoc_single(vert, norm, om, se3.identity(), T_fm, dist_thr=0.05)
    n = kat1_reference_asserts(vert, dbg["da"], dbg["err"])
    assert n == len(pts) and n > 10


def _fd_check(name, mode):
    m = scenes.articulated(name)
    om = O.OModel(m)
    d = 1e-3
    rng = np.random.default_rng(7)
    worst = 0.0
    for n in range(6):
        joints = scenes.random_joints(m, 100 + n, margin=d)
        T_mf, T_fm = om.fk(joints, mode)
        for frame in range(m.n_frames):
            for pf in (np.array([0, 0, 0], dtype=F), np.array([0.1, 0.1, 0.1], dtype=F)):
                pm = se3.apply(T_mf[frame], pf)
                J3 = O.model_jacobian_3d(om, T_fm, T_mf, pm, frame)
                comps = [O.error_jacobian(om, T_fm, T_mf, pm, frame, e) for e in np.eye(3, dtype=F)]
                for j in range(m.n_joints):
                    jp, jm = joints.copy(), joints.copy()
                    jp[j] += d
                    jm[j] -= d
                    pos = se3.apply(om.fk(jp, mode)[0][frame], pf)
                    neg = se3.apply(om.fk(jm, mode)[0][frame], pf)
                    num = (pos.astype(np.float64) - neg.astype(np.float64)) / (2 * d)
                    # CPU test compares magnitudes (test_model_jacobian.cpp:70-73), the GPU test components (.cu)
                    assert abs(np.linalg.norm(num) - np.linalg.norm(J3[6 + j])) < 1e-4
                    for k in range(3):
                        assert abs(num[k] - comps[k][6 + j]) < 1e-4
                        worst = max(worst, abs(num[k] - comps[k][6 + j]))
                # 6-DoF part: d(point_m)/d(twist) with T_mc <- exp(dt) T_mc moves model points by -(v + w x p)
                for k in range(3):
                    assert comps[k][k] == -1 and comps[k][(k + 1) % 3] == 0
    return worst


def test_kat3_jacobian_finite_difference_left_hand_axis_angle():
    assert _fd_check("left_hand", "axis") < 1e-4


def test_kat3_jacobian_finite_difference_left_hand_dh():
    # leftHand has unit single-axis joints and no DH offsets: both FK conventions agree (Appendix A-11)
    assert _fd_check("left_hand", "dh") < 1e-4


def test_kat3_jacobian_finite_difference_allegro():
    assert _fd_check("allegro", "dh") < 1e-4


def test_kat3_6dof_jacobian_matches_twist_derivative():
    m = scenes.articulated("left_hand")
    om = O.OModel(m)
    d = 1e-3
    for n in range(5):
        joints = scenes.random_joints(m, 300 + n)
        T_mf, T_fm = om.fk(joints, "dh")
        rng = np.random.default_rng(n)
        t6 = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-np.pi, np.pi, 3)]).astype(F)
        T_mc = O.se3_exp(t6)
        T_cm = O.se3_invert(T_mc)
        for frame in (0, 3, 9, m.n_frames - 1):
            pf = np.array([0.1, 0.1, 0.1], dtype=F)
            pm = se3.apply(T_mf[frame], pf)
            J3 = O.model_jacobian_3d(om, T_fm, T_mf, pm, frame)
            for i in range(6):
                dt = np.zeros(6, dtype=F)
                dt[i] = -d
                neg = se3.apply(O.se3_invert(O.se3_mul(O.se3_exp(dt), T_mc)), pm)
                dt[i] = d
                pos = se3.apply(O.se3_invert(O.se3_mul(O.se3_exp(dt), T_mc)), pm)
                num_c = (pos.astype(np.float64) - neg.astype(np.float64)) / (2 * d)
                num_m = se3.rotate(T_mc, num_c.astype(F))
                assert abs(np.linalg.norm(num_m) - np.linalg.norm(J3[i])) < 1e-4 * max(1.0, np.linalg.norm(J3[i]))


def test_kat4_null_pose_reduction():
    n_dims = 10
    n_art = n_dims - 6
    mins = [-float(i) for i in range(n_art)]
    maxs = [float(i) for i in range(n_art)]
    names = [str(i) for i in range(n_art)]
    red = NullReduction(n_art, mins, maxs, names)
    assert red.isNull()
    pose = Pose(red)
    assert not pose.isReduced()
    assert pose.getDimensions() == n_dims
    assert pose.getArticulatedDimensions() == n_art
    assert pose.getReducedDimensions() == n_dims
    assert pose.getReducedArticulatedDimensions() == n_art
    for i in range(n_art):
        assert pose.getReducedMin(i) == -float(i)
        assert pose.getReducedMax(i) == float(i)
    pose.getReducedArticulation()[0] = 5.0
    assert pose.getArticulation()[0] == 5.0
    pose.getArticulation()[n_art - 1] = F(4.2)
    assert pose.getReducedArticulation()[n_art - 1] == F(4.2)
