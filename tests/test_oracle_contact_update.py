"""CPU tests of the oracle's update_contact_point path (a17): projectToSdfSurface (sdf.cpp:6-23), ContactPrior::updatePriorParams
(contact_prior.cpp:207-221) and the prior-parameter columns of the solve (optimizer.cpp:829-832, 913-958).

Pins (the reference holds no fixture for this path -- "parity unpinned" beyond these):
  * the projection against the analytic sphere it must land on;
  * the loop's treatment of the prior unknowns against an independent assembly here of the reference's FULL system -- three
    unknowns per prior, frozen or not, exactly as optimizer.cpp:829-832 lays it out -- solved in fp64 with Eigen's column order."""
import numpy as np

import oracle_lib as O
from dart_amd import configs, se3
from dart_amd.model import sphere_model

from independent_ne import trilinear64

F = np.float32


def test_projection_lands_on_the_analytic_sphere():
    hm = sphere_model(0.1).voxelize(0.005, 0.05)       # radius 10 cm, 5 mm voxels, 5 cm padding
    om = O.OModel(hm, 0)
    g = om.grid(0)
    sdf = hm.sdfs[0]
    off, res = np.asarray(sdf.offset, dtype=np.float64), float(sdf.resolution)
    rng = np.random.default_rng(5)
    for _ in range(20):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        p = d * rng.uniform(0.06, 0.135)               # inside and outside the surface
        pg = O.project_to_sdf_surface(g, (p - off) / res, 1e-9, 100)
        pw = pg.astype(np.float64) * res + off
        # the model SDFs are distance transforms of the voxelised sphere (voxel units): their zero level set lies within a voxel
        # of the analytic surface, and the projection must sit ON that level set (independent fp64 interpolation)
        assert abs(np.linalg.norm(pw) - 0.1) < 1.0 * res
        print("projection residual (voxels):", float(trilinear64(sdf, pg[None].astype(np.float64))[0]))
        assert abs(trilinear64(sdf, pg[None].astype(np.float64))[0]) < 0.02
        assert np.dot(pw / np.linalg.norm(pw), d) > 0.999          # moved along the radius
    # a loose threshold stops at once; a point outside the gradient-interpolation bounds is left alone (sdf.cpp:10-11)
    p0 = (np.array([0.12, 0.0, 0.0]) - off) / res
    assert np.array_equal(O.project_to_sdf_surface(g, p0, 1e3, 100), p0.astype(F))
    out = np.array([1.0, 1.0, 1.0], dtype=F)
    assert np.array_equal(O.project_to_sdf_surface(g, out, 1e-9, 100), out)
    # one iteration = one step of the distance along the normalised gradient
    one = O.project_to_sdf_surface(g, p0, 1e-9, 1)
    v0 = trilinear64(sdf, p0[None])[0]
    assert abs((p0[0] - one[0]) - v0) < 1e-3 and 3.4 < v0 < 4.6 and abs(one[1] - p0[1]) < 1e-3


def scene(n_update=3):
    sc = configs.contact_grasp_scene(1, n_attract=24, n_repulse=40, iterations=1)
    hand, obj = sc.hand, sc.obj
    oms = [O.OModel(hand, 0), O.OModel(obj, 1)]
    priors = list(sc.priors)
    thumb = priors[-1]
    thumb.update_contact_point = True                   # hand -> object prior whose contact point slides over the thumb tip
    # more fingertip priors of the same kind: points on other hand SDFs' surfaces near the object
    from dart_amd.tracker_no_obs import ContactPrior
    T_mf = hand.forward_kinematics(sc.init_joints[0], "dh")
    T_cm = se3.invert(sc.init_T_mc[0])
    added = 0
    cloud = hand.collision_cloud(2e4, seed=3)
    a = np.asarray(configs.OBJ_RADII, dtype=np.float64)
    for s in np.unique(cloud[:, 3].astype(int)):
        if added >= n_update - 1 or s == thumb.srcSdfNum:
            continue
        pts = cloud[cloud[:, 3].astype(int) == s, :3]
        pc = se3.apply(T_cm, se3.apply(T_mf[hand.sdf_frame[int(s)]], pts)).astype(np.float64)
        lvl = np.abs(np.sqrt(((pc / a) ** 2).sum(1)) - 1.0) * a.min()
        k = int(np.argmin(lvl))
        if lvl[k] < 0.012:
            priors.append(ContactPrior(0, 1, int(s), 0, 25.0, tuple(pts[k]), True, True, True, False, 0.02))
            added += 1
    assert sum(p.update_contact_point for p in priors) >= 2
    return sc, oms, priors


def full_reference_system(oms, priors, T_mcs, joints, reg, lm):
    """The reference's own layout: model blocks, then three unknowns for EVERY prior (optimizer.cpp:829-832); contributions in
    list order, regularisation + LM damping of the model diagonals after the priors (:913-929); back substitution on the upper
    triangle, last column first, a column skipped when its right-hand side is exactly 0 (Eigen's sparse column-major solve)."""
    n = len(oms)
    dims = [o.dims for o in oms]
    offs = np.concatenate([[0], np.cumsum(dims)])[:n]
    D = int(sum(dims))
    sysN = D + 3 * len(priors)
    JTJ, JTe = np.zeros((sysN, sysN), dtype=F), np.zeros(sysN, dtype=F)
    T_cms, T_fms, T_mfs = [], [], []
    for m in range(n):
        T_mf, T_fm = oms[m].fk(joints[m], "dh")
        T_cms.append(O.se3_invert(T_mcs[m])); T_fms.append(T_fm); T_mfs.append(T_mf)
    errs = []
    for k, p in enumerate(priors):
        e, _ = O.contact_prior_contribution(oms, p, T_mcs, T_cms, T_fms, T_mfs, offs, dims, D + 3 * k, JTJ, JTe)
        errs.append(e)
    for m in range(n):
        for i in range(dims[m]):
            d = offs[m] + i
            JTJ[d, d] = F(JTJ[d, d] + F(reg))
            JTJ[d, d] = F(JTJ[d, d] + F(F(lm) * JTJ[d, d]))
    x = JTe.astype(np.float64)
    U = JTJ.astype(np.float64)
    for i in range(sysN - 1, -1, -1):
        if x[i] != 0.0:
            x[i] /= U[i, i]
            nz = np.nonzero(U[:i, i])[0]
            x[nz] -= x[i] * U[nz, i]
    return -x, D, errs


def test_loop_matches_the_full_prior_parameter_system():
    sc, oms, priors = scene()
    T0, j0 = [sc.init_T_mc[0], se3.identity()], [sc.init_joints[0], np.zeros(0, dtype=F)]
    from dart_amd import _abi
    o = _abi.NoObsOptions()
    o.numIterations, o.nModels = 1, 2
    reg, lm = 1e-4, 25.0
    for m in range(2):
        o.regularization[m], o.regularizationScaled[m] = reg, lm
    clouds = [np.zeros((0, 4), dtype=F)] * 2
    pots = [np.ones(len(oms[m].host.sdfs) ** 2, dtype=np.int32) for m in range(2)]
    with O.contact_out(len(priors)) as co:
        oT, oj, oe, info, first = O.optimize_poses_noobs(oms, T0, j0, o, clouds, pots, priors)
        cps = co.points.copy()
    x, D, errs = full_reference_system(oms, priors, T0, j0, reg, lm)
    # poses: paramUpdate applied as optimizer.cpp:933-954 does
    nj = oms[0].n_joints
    T_new = O.se3_mul(O.se3_exp(x[0:6].astype(F)), T0[0])
    lim = np.asarray(sc.hand.joint_limits, dtype=F).reshape(-1, 2)
    j_new = np.clip(j0[0] + x[6:6 + nj].astype(F), lim[:, 0], lim[:, 1])
    assert np.abs(oT[0] - T_new).max() < 1e-5 and np.abs(oj[0] - j_new).max() < 1e-5
    assert np.abs(oT[0] - T0[0]).max() > 1e-4                      # the hand did move
    # contact points: frozen ones untouched, updating ones stepped by their three unknowns and projected to their SDF's surface
    n_moved = 0
    for k, p in enumerate(priors):
        if not p.update_contact_point:
            assert np.array_equal(cps[k], np.asarray(p.contactPoint, dtype=F))
            assert not x[D + 3 * k:D + 3 * k + 3].any()           # their rows never receive a non-zero (SURVEY A-7)
            continue
        ref = O.contact_prior_update_params(oms, p, x[D + 3 * k:D + 3 * k + 3].astype(F))
        assert np.abs(cps[k] - ref).max() < 1e-6, (k, cps[k], ref)
        g = oms[p.srcModelID].grid(p.srcSdfNum)
        sdf = oms[p.srcModelID].host.sdfs[p.srcSdfNum]
        pg = (cps[k].astype(np.float64) - np.asarray(sdf.offset)) / float(sdf.resolution)
        inb = all(2.51 < pg[a] < sdf.dim[a] - 2.51 for a in range(3))
        if inb:
            assert abs(trilinear64(sdf, pg[None])[0]) < 1e-3       # on the zero level set (voxel units)
        n_moved += int(np.abs(cps[k] - np.asarray(p.contactPoint, dtype=F)).max() > 1e-6)
    assert n_moved >= 2
    # an active updating prior pulls its own contact point, and its pull on the models cancels in the back substitution: solving
    # with those priors frozen instead moves the hand differently
    frozen = [type(p)(**{**p.__dict__, "update_contact_point": False}) for p in priors]
    fT, fj, _, _, _ = O.optimize_poses_noobs(oms, T0, j0, o, clouds, pots, frozen)
    assert np.abs(fT[0] - oT[0]).max() > 1e-5
    # energy after the iteration is evaluated at the moved contact points (the reference mutates the prior objects)
    moved = [type(p)(**{**p.__dict__, "contactPoint": tuple(cps[k])}) for k, p in enumerate(priors)]
    e_moved = O.get_error_noobs(oms, oT, oj, o, clouds, pots, moved)
    assert abs(e_moved - oe) < 1e-6 * max(1.0, abs(oe))
