"""GPU parity: ContactGrasp refinement (TrackerNoObs::optimizePoses -> OptimizerNoObs::optimizePoses / getError with
ContactPrior::computeContribution) vs the CPU oracle's restatement of optimizer.cpp:783-1076 + contact_prior.cpp:8-205.
No test of the reference covers this path: parity is pinned by the restatement only ("unpinned")."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dart_amd import _abi, configs, se3
from dart_amd.tracker_no_obs import ContactPrior
from test_gpu_optimize_pose import pose_diff

F = np.float32
pytestmark = pytest.mark.gpu

_scene_cache = {}


def scene(n):
    if n not in _scene_cache:
        _scene_cache[n] = configs.contact_grasp_scene(n)
    return _scene_cache[n]


def oracle_run(sc, tr, T_hand, joints_hand, iters):
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    tr.getOptions().numIterations = iters
    o = tr._options_struct()
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    return O.optimize_poses_noobs(oms, [T_hand, se3.identity()], [joints_hand, np.zeros(0, dtype=F)], o, clouds, pots, sc.priors)


def sys_rel_err(got, ref, D):
    """norm-relative error of the upper JTJ block and of JTe (first-iteration system, before regularisation)."""
    out = 0.0
    for sl in (slice(0, D * D), slice(D * D, D * D + D)):
        n = np.linalg.norm(ref[sl].astype(np.float64))
        out = max(out, np.linalg.norm(got[sl].astype(np.float64) - ref[sl].astype(np.float64)) / (n if n > 0 else 1.0))
    return out


def test_first_system_energy_and_short_run_match_oracle(gpu):
    sc = scene(4)
    tr = configs.make_grasp_tracker(sc)
    T, j = configs.grasp_batch_arrays(sc)
    D = 32
    for c in (0, 1, 3):
        tr.getOptions().numIterations = 3
        gT, gj, ge, ginfo, gfirst = tr.optimizePosesBatch(T[c:c + 1], j[c:c + 1], debug=True)
        oT, oj, oe, oinfo, ofirst = oracle_run(sc, tr, T[c, 0], j[c], 3)
        assert np.count_nonzero(ofirst) > 50
        assert sys_rel_err(gfirst[0], ofirst, D) < 1e-4
        # per-iteration intersection error and prior energy
        assert np.allclose(ginfo[0, :, 0], oinfo[:, 0], rtol=1e-4, atol=1e-7)
        # the reference (and the oracle) add the 2501 signed prior errors (|e| up to w*thr = 3, sum of |e| ~ 2e3) one by
        # one in fp32; the device sums tiles in fp32 and tiles in fp64: agreement is bounded by fp32 rounding of that sum
        assert np.allclose(ginfo[0, :, 1], oinfo[:, 1], rtol=2e-4, atol=5e-2)
        dt, dr = pose_diff(gT[0, 0], oT[0])
        dto, dro = pose_diff(gT[0, 1], oT[1])
        dj = float(np.abs(gj[0] - oj[0]).max())
        print("candidate %d after 3 iterations: hand |dt|=%.2e |dr|=%.2e max|dq|=%.2e  object |dt|=%.2e |dr|=%.2e  energy gpu %.6f oracle %.6f"
              % (c, dt, dr, dj, dto, dro, ge[0], oe))
        assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3 and dto < 1e-3 and dro < 1e-3
        assert abs(ge[0] - oe) < 1e-3 * max(1.0, abs(oe))


def test_get_error_matches_oracle(gpu):
    sc = scene(4)
    tr = configs.make_grasp_tracker(sc)
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    o = tr._options_struct()
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    T, j = configs.grasp_batch_arrays(sc)
    T[3, 0] = sc.gt_T_mc
    j[3] = sc.gt_joints
    e = tr.getError(T, j)
    for c in range(4):
        ref = O.get_error_noobs(oms, [T[c, 0], T[c, 1]], [j[c], np.zeros(0, dtype=F)], o, clouds, pots, sc.priors)
        assert abs(e[c] - ref) < 1e-4 * max(1.0, abs(ref)), (c, e[c], ref)
    # tracker-level call on the current poses
    tr.setPose(0, sc.gt_T_mc, sc.gt_joints)
    assert abs(tr.getError() - e[3]) < 1e-6


def segmented_grasp(name, sc, tr, T1, j1, iters):
    """tests/trajectory.py for one grasp candidate (T1 [2,12], j1 [nJ]): GPU prefixes (batch of one) vs the oracle restarted
    from the GPU's poses at every decision flip. Returns (segments, gpu end pose, gpu energy, oracle energy of the last segment)."""
    from trajectory import parts, segmented_parity
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    energies, poses = {}, {}

    def gpu_prefix(k):
        tr.getOptions().numIterations = k
        gT, gj, ge, ginfo, _ = tr.optimizePosesBatch(T1[None], j1[None], debug=True)
        energies[k] = float(ge[0])
        poses[k] = (gT[0, 0].copy(), gT[0, 1].copy(), gj[0].copy())
        return poses[k], ginfo[0].copy()

    def oracle_from(pose, k0, n):
        tr.getOptions().numIterations = n
        o = tr._options_struct()
        oT, oj, oe, oinfo, _ = O.optimize_poses_noobs(oms, [pose[0], pose[1]], [pose[2], np.zeros(0, dtype=F)], o, clouds, pots, sc.priors)
        energies[("oracle", k0 + n)] = oe
        return (oT[0], oT[1], oj[0]), oinfo

    def decisions_equal(g, o):
        # sums of ~2500 signed prior errors (fp32 rounding of the oracle's sequential sum, see above) and the intersection error
        return bool(abs(g[1] - o[1]) <= 5e-2 + 2e-4 * abs(o[1]) and abs(g[0] - o[0]) <= 1e-4 * abs(o[0]) + 1e-7)

    def pose_err(a, b):
        dt, dr = pose_diff(a[0], b[0])
        dto, dro = pose_diff(a[1], b[1])
        return parts(translation_m=[dt, dto], rotation_rad=[dr, dro], joints_rad=[np.abs(a[2] - b[2]).max()])

    def near_ties(gpose, opose, k):
        import near_ties as NT
        dims = [oms[0].dims, oms[1].dims]
        zero = np.zeros(0, dtype=F)
        recs = NT.prior_near_ties(oms, sc.priors, ([gpose[0], gpose[1]], [gpose[2], zero]), ([opose[0], opose[1]], [opose[2], zero]),
                                  dims, [0, dims[0]])
        return recs

    segs = segmented_parity(name, iters, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3, near_ties=near_ties)
    tr.getOptions().numIterations = iters
    return segs, poses[iters], energies[iters], energies[("oracle", iters)]


def test_full_refinement_40_iterations(gpu):
    """BASELINE config 3: 40 iterations (grasp_analyzer.cpp:198). Every segment of the trajectory between decision flips
    (a contact prior crossing its threshold / cylinder gate, a collision point entering or leaving an SDF) is held to 1e-3
    (tests/trajectory.py: the oracle is restarted from the GPU's poses at every flip)."""
    sc = scene(4)
    tr = configs.make_grasp_tracker(sc)
    T, j = configs.grasp_batch_arrays(sc)
    segs, _, ge, oe = segmented_grasp("C3 ContactGrasp 40 iterations", sc, tr, T[0], j[0], 40)
    print("energy after 40 iterations: gpu %.6f, oracle (last segment) %.6f" % (ge, oe))
    assert abs(ge - oe) < 1e-3 * max(1.0, abs(oe))
    assert segs[0][1] >= 10, "the first decision flip came at iteration %d" % segs[0][1]


def test_batch_equals_single_and_is_order_independent(gpu):
    """Sharding property (SURVEY 7 test plan vi): a candidate's result does not depend on the batch it is in."""
    sc = scene(16)
    tr = configs.make_grasp_tracker(sc, max_batch=16)
    T, j = configs.grasp_batch_arrays(sc)
    tr.getOptions().numIterations = 6
    bT, bj, be, _, _ = tr.optimizePosesBatch(T, j)
    perm = np.random.default_rng(0).permutation(16)
    pT, pj, pe, _, _ = tr.optimizePosesBatch(T[perm], j[perm])
    assert np.array_equal(pT, bT[perm]) and np.array_equal(pj, bj[perm]) and np.array_equal(pe, be[perm])
    for c in (0, 5, 15):
        sT, sj, se_, _, _ = tr.optimizePosesBatch(T[c:c + 1], j[c:c + 1])
        # a large batch uses fewer, longer workgroups per term than a single candidate (different fixed summation
        # order), so the comparison is to tolerance, not bitwise
        assert np.abs(sT[0] - bT[c]).max() < 1e-4 and np.abs(sj[0] - bj[c]).max() < 1e-4
    assert np.isfinite(be).all()


def test_edge_cases(gpu):
    sc = scene(4)
    T, j = configs.grasp_batch_arrays(sc)
    # zero iterations: poses untouched, energy = getError
    tr = configs.make_grasp_tracker(sc)
    tr.getOptions().numIterations = 0
    gT, gj, ge, _, _ = tr.optimizePosesBatch(T[:2], j[:2])
    lim = np.asarray(sc.hand.joint_limits, dtype=F).reshape(-1, 2)
    assert np.array_equal(gT, T[:2]) and np.array_equal(gj, np.clip(j[:2], lim[:, 0], lim[:, 1]))
    assert np.allclose(ge, tr.getError(T[:2], j[:2]), rtol=1e-6)
    # zero-weight priors contribute nothing; priors far from everything return sign*w*threshold (contact_prior.cpp:18)
    far = [ContactPrior(1, 0, 0, -1, 2.0, (5.0, 5.0, 5.0), False, True, True, False, 0.02),
           ContactPrior(1, 0, 0, -1, 3.0, (5.0, 5.0, 5.0), False, True, True, True, 0.02),
           ContactPrior(1, 0, 0, -1, 0.0, (0.0, 0.0, 0.06), False, True, True, False, 0.02)]
    sc2 = configs.contact_grasp_scene(1)
    sc2.priors = far
    tr2 = configs.make_grasp_tracker(sc2)
    o = tr2.getOptions()
    n = 2
    o.lambdaIntersection = [0.0] * (n * n)
    T1, j1 = configs.grasp_batch_arrays(sc2)
    e = tr2.getError(T1, j1)
    assert abs(e - (2.0 * 0.02 - 3.0 * 0.02 + 0.0) / 3.0) < 1e-7
    # a prior between a model and itself is rejected loudly (its source and destination blocks would alias)
    sc2.priors = [ContactPrior(0, 0, 0, -1, 1.0, (0, 0, 0.06), False, True, True, False, 0.02)]
    tr3 = configs.make_grasp_tracker(sc2)
    with pytest.raises(_abi.DartkError):
        tr3.getError(T1, j1)
    # update_contact_point: getError alone never moves a contact point (tests/test_gpu_contact_update.py covers the moves)
    sc2.priors = [ContactPrior(1, 0, 0, -1, 1.0, (0, 0, 0.06), True, True, True, False, 0.02)]
    tr4 = configs.make_grasp_tracker(sc2)
    o4 = tr4.getOptions()
    o4.lambdaIntersection = [0.0] * (n * n)
    sc2.priors = [ContactPrior(1, 0, 0, -1, 1.0, (0, 0, 0.06), False, True, True, False, 0.02)]
    tr5 = configs.make_grasp_tracker(sc2)
    tr5.getOptions().lambdaIntersection = [0.0] * (n * n)
    assert tr4.getError(T1, j1) == tr5.getError(T1, j1)


def test_full_size_batch_properties(gpu):
    """BASELINE config 4 at its full size (4096 candidates x 2501 priors), through properties that need no oracle run of
    the whole batch: (i) a sample of candidates against the oracle, (ii) reversing the batch reverses the result bit for
    bit, (iii) duplicated candidates give duplicated results, (iv) getError at the refined poses is the returned energy."""
    n = 4096
    sc = scene(n)
    tr = configs.make_grasp_tracker(sc, max_batch=n)
    T, j = configs.grasp_batch_arrays(sc)
    T, j = T.copy(), j.copy()
    T[n - 7], j[n - 7] = T[11], j[11]           # duplicates far apart in the batch (different workgroups, different XCDs)
    T[2048], j[2048] = T[11], j[11]
    iters = 40                                  # the configuration's own iteration count (grasp_analyzer.cpp:198)
    tr.getOptions().numIterations = iters
    gT, gj, ge, _, _ = tr.optimizePosesBatch(T, j)
    assert np.isfinite(gT).all() and np.isfinite(gj).all() and np.isfinite(ge).all()
    for c in (n - 7, 2048):
        assert np.array_equal(gT[c], gT[11]) and np.array_equal(gj[c], gj[11]) and ge[c] == ge[11]
    rT, rj, re_, _, _ = tr.optimizePosesBatch(T[::-1].copy(), j[::-1].copy())
    assert np.array_equal(rT[::-1], gT) and np.array_equal(rj[::-1], gj) and np.array_equal(re_[::-1], ge)
    # a sample of candidates over all 40 iterations: the candidate alone (batch of one: other workgroup split, other fixed
    # summation order) against the oracle segment by segment, and the batch's result against the candidate alone
    tr1 = configs.make_grasp_tracker(sc, max_batch=1)
    for c in (0, 11, 777, 1500, 2600, 3333, n - 1):
        segs, pose1, e1, oe = segmented_grasp("C4 candidate %d of %d" % (c, n), sc, tr1, T[c], j[c], iters)
        dt, dr = pose_diff(gT[c, 0], pose1[0])
        dto, dro = pose_diff(gT[c, 1], pose1[1])
        dj = float(np.abs(gj[c] - pose1[2]).max())
        print("candidate %4d: batch vs alone |dt|=%.2e |dr|=%.2e max|dq|=%.2e; energy batch %.6f alone %.6f oracle %.6f" % (c, dt, dr, dj, ge[c], e1, oe))
        assert max(dt, dr, dto, dro, dj) < 1e-3
        assert abs(e1 - oe) < 1e-3 * max(1.0, abs(oe)) and abs(ge[c] - e1) < 1e-3 * max(1.0, abs(e1))
    e2 = tr.getError(gT, gj)
    assert np.allclose(e2, ge, rtol=1e-6, atol=1e-7)
    # the refinement lowers the mean energy of the batch
    e0 = tr.getError(T, j)
    assert float(np.mean(ge)) < float(np.mean(e0))
