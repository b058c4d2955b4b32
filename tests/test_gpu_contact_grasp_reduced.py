"""GPU parity: ContactGrasp refinement with a REDUCED pose -- the reference's own Barrett set-up (grasp_analyzer.cpp:393-420: the
Barrett hand's 8 joints driven by a LinearPoseReduction with 4 parameters) through TrackerNoObs / OptimizerNoObs
(optimizer.cpp:783-983): reduced intersection terms, contact priors indexed by the reduced dimensions (contact_prior.cpp:31-32),
the update on the reduced articulation with the reduction's limits and the projection to the joints, vs the CPU oracle."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

import oracle_lib as O
from dart_amd import configs, se3
from dart_amd.pose import ParamMapPoseReduction
from dart_amd.tracker_no_obs import ContactPrior, TrackerNoObs
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu


def barrett_scene(reduction_kind="linear", n_attract=120, n_repulse=300):
    """Barrett hand (camera frame = model frame) and the ellipsoid object placed among its fingers: object surface points
    within 2 cm of the posed hand are contact (attract) points, the rest repulse; hand self- and hand<->object intersection."""
    sc = configs.config5(160, 90, 0)
    hand, obj = sc.hand, sc.obj
    red = sc.reduction
    if reduction_kind == "map":
        red = ParamMapPoseReduction(8, 4, [0, 1, 1, 0, 2, 2, 3, 3], [red.getMin(i) for i in range(4)], [red.getMax(i) for i in range(4)],
                                    hand.joint_names[:4])
    rng = np.random.default_rng(77)
    art = np.array([0.2, 0.9, 0.9, 0.9], dtype=F)
    full = np.zeros(8, dtype=F)
    red.projectReducedToFull(art, full)
    cloud = hand.collision_cloud(2e4, seed=3)
    T_mf = hand.forward_kinematics(full, "dh")
    pts_m = np.zeros((len(cloud), 3), dtype=F)
    for sn in np.unique(cloud[:, 3].astype(int)):
        sel = cloud[:, 3].astype(int) == sn
        pts_m[sel] = se3.apply(T_mf[hand.sdf_frame[int(sn)]], cloud[sel, :3])
    centre = pts_m.mean(0) + np.array([0.0, 0.0, 0.02], dtype=F)
    T_mc_obj = se3.translation(*(-centre))                      # camera (= hand model frame) -> object model frame
    a = np.asarray(configs.OBJ_RADII, dtype=np.float64)
    v = rng.normal(size=(4000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    surf = v * a
    nrm = surf / a ** 2; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d, _ = cKDTree(pts_m.astype(np.float64)).query(surf + centre)
    contact, free = np.where(d < 0.02)[0][:n_attract], np.where(d >= 0.02)[0][:n_repulse]
    assert len(contact) > 20 and len(free) > 50
    priors = [ContactPrior(1, 0, 0, -1, 150.0, tuple(surf[i].astype(F)), False, True, True, False, 0.02, tuple(nrm[i].astype(F))) for i in contact]
    priors += [ContactPrior(1, 0, 0, -1, 20.0, tuple(surf[i].astype(F)), False, True, True, True, 0.02, tuple(nrm[i].astype(F))) for i in free]
    tr = TrackerNoObs(max_batch=4)
    h = tr.addModel(hand, cloud_seed=21, reduction=red)
    tr.setIntersectionPotentialMatrix(h, sc.potential)
    ob = tr.addModel(obj, cloud_seed=22)
    for p in priors:
        tr.addPrior(p)
    o = tr.getOptions()
    n = 2
    o.lambdaIntersection[h + n * h] = 5.0
    o.lambdaIntersection[ob + n * h] = 100.0
    o.lambdaIntersection[h + n * ob] = 100.0
    o.regularization = [10 ** -1.0] * n             # log10 reg -1 for Barrett (grasp_analyzer.cpp:219-224)
    o.regularizationScaled = [25.0] * n
    # candidates: perturbed hand poses around the nominal one (reduced articulations)
    B = 4
    T = np.zeros((B, 2, 12), dtype=F)
    J = np.zeros((B, 4), dtype=F)
    for c in range(B):
        t6 = np.concatenate([rng.uniform(-0.01, 0.01, 3), rng.uniform(-0.08, 0.08, 3)]).astype(F)
        T[c, 0] = se3.mul(se3.exp(t6), se3.identity())
        T[c, 1] = T_mc_obj
        J[c] = np.clip(art + rng.uniform(-0.15, 0.15, 4).astype(F), [red.getMin(i) for i in range(4)], [red.getMax(i) for i in range(4)])
    return sc, tr, priors, T, J


def oracle_args(sc, tr):
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    return oms, clouds, pots


@pytest.mark.parametrize("kind", ["linear", "map"])
def test_reduced_refinement_matches_oracle(gpu, kind):
    from trajectory import parts, segmented_parity
    sc, tr, priors, T, J = barrett_scene(kind)
    oms, clouds, pots = oracle_args(sc, tr)
    dims = [tr.reduced_dims(0), tr.reduced_dims(1)]
    assert dims == [10, 6]
    D = 16
    iters = 12
    with O.noobs_reductions(tr):
        # energies at the start poses
        e = tr.getError(T, J)
        o = tr._options_struct()
        for c in range(len(T)):
            ref = O.get_error_noobs(oms, [T[c, 0], T[c, 1]], [J[c], np.zeros(0, dtype=F)], o, clouds, pots, priors)
            assert abs(e[c] - ref) < 1e-4 * max(1.0, abs(ref)), (c, e[c], ref)
        # first system (reduced dimensions) and the whole trajectory, candidate by candidate
        for c in (0, 2):
            tr.getOptions().numIterations = 2
            gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(T[c:c + 1], J[c:c + 1], debug=True)
            o = tr._options_struct()
            oT, oj, oe, oinfo, ofirst = O.optimize_poses_noobs(oms, [T[c, 0], T[c, 1]], [J[c], np.zeros(0, dtype=F)], o, clouds, pots, priors, dims=dims)
            assert gfirst.shape[1] == D * D + D and np.count_nonzero(ofirst) > 60
            for sl in (slice(0, D * D), slice(D * D, D * D + D)):
                nref = np.linalg.norm(ofirst[sl].astype(np.float64))
                assert np.linalg.norm(gfirst[0][sl].astype(np.float64) - ofirst[sl]) / nref < 1e-4
            assert abs(ginfo[0, 0, 0] - oinfo[0, 0]) <= 1e-4 * abs(oinfo[0, 0]) + 1e-7 and oinfo[0, 0] > 0       # intersections are active

            def gpu_prefix(k):
                tr.getOptions().numIterations = k
                gT, gj, ge, ginfo, _ = tr.optimizePosesBatch(T[c:c + 1], J[c:c + 1], debug=True)
                return (gT[0, 0].copy(), gT[0, 1].copy(), gj[0].copy()), ginfo[0].copy()

            def oracle_from(pose, k0, n):
                tr.getOptions().numIterations = n
                o_ = tr._options_struct()
                oT, oj, oe, oinfo, _ = O.optimize_poses_noobs(oms, [pose[0], pose[1]], [pose[2], np.zeros(0, dtype=F)], o_, clouds, pots, priors, dims=dims)
                return (oT[0], oT[1], oj[0]), oinfo

            def decisions_equal(g, o_):
                return bool(abs(g[1] - o_[1]) <= 5e-2 + 2e-4 * abs(o_[1]) and abs(g[0] - o_[0]) <= 1e-4 * abs(o_[0]) + 1e-7)

            def pose_err(a, b):
                dt, dr = pose_diff(a[0], b[0])
                dto, dro = pose_diff(a[1], b[1])
                return parts(translation_m=[dt, dto], rotation_rad=[dr, dro], joints_rad=[np.abs(a[2] - b[2]).max()])

            segmented_parity("Barrett ContactGrasp (%s reduction) candidate %d" % (kind, c), iters, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3)
            end, _ = gpu_prefix(iters)
            assert float(np.abs(end[2] - J[c]).max()) > 1e-4 or pose_diff(end[0], T[c, 0])[0] > 1e-5      # the refinement moved the hand
            red = tr.reductions[0]
            assert all(red.getMin(i) - 1e-6 <= end[2][i] <= red.getMax(i) + 1e-6 for i in range(4))
    # batch == singles
    tr.getOptions().numIterations = 5
    bT, bj, be, _, _ = tr.optimizePosesBatch(T, J)
    for c in range(len(T)):
        sT, sj, se_, _, _ = tr.optimizePosesBatch(T[c:c + 1], J[c:c + 1])
        assert np.abs(sT[0] - bT[c]).max() < 1e-4 and np.abs(sj[0] - bj[c]).max() < 1e-4


@pytest.mark.parametrize("switch", ["DARTK_NOOBS_NO_MFMA", "DARTK_NOOBS_NO_TILE_CULL"])
def test_contraction_and_culling_variants_are_bit_identical(gpu, switch, monkeypatch):
    """Systems of <= 32 dimensions are contracted with v_mfma_f32_32x32x2_f32 (an fp32 fma chain in row order, the same chain the
    VALU register tiles run) and the self-intersection term skips (tile, SDF) pairs whose bounding spheres are apart: both are
    pure accelerations, so the first system, the per-iteration records and the end poses must not change by a single bit when the
    diagnostic switches turn them off. (The culling switch is read when a model is added: a second tracker is built under it.)"""
    def run():
        sc, tr, priors, T, J = barrett_scene("linear")
        tr.getOptions().numIterations = 6
        return tr.optimizePosesBatch(T, J, debug=True)
    a = run()
    monkeypatch.setenv(switch, "1")
    b = run()
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        assert x.shape == y.shape and x.tobytes() == y.tobytes()
    assert np.count_nonzero(a[4]) > 60
