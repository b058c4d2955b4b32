"""CPU tests of ContactGrasp's file formats and of the point selection (dart_amd/grasp_analyzer.py)."""
import json
import os

import numpy as np

from dart_amd import grasp_analyzer as GA

HERE = os.path.dirname(os.path.abspath(__file__))


def _fnv(idx):
    h = 1469598103934665603
    for v in idx:
        h ^= v
        h = (h * 1099511628211) % (1 << 64)
    return str(h)


def test_std_shuffle_matches_libstdcxx_golden():
    """tests/golden/std_shuffle.json was produced by g++'s std::shuffle + std::default_random_engine (gen_std_shuffle.cpp):
    two shuffles in a row with one engine, including list sizes on both sides of the two-swaps-per-draw switch."""
    cases = json.load(open(os.path.join(HERE, "golden", "std_shuffle.json")))
    assert len(cases) >= 5
    for c in cases:
        rng = GA.MinStdRand0()
        a = GA.std_shuffle(list(range(c["sizes"][0])), rng)
        b = GA.std_shuffle(list(range(c["sizes"][1])), rng)
        assert a[:40] == c["first"] and _fnv(a) == c["hash0"], c["sizes"]
        assert b[:40] == c["second"] and _fnv(b) == c["hash1"], c["sizes"]


def test_grasp_csv_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    T = rng.normal(size=(5, 12)).astype(np.float32)
    J = rng.uniform(-1, 1, size=(5, 20)).astype(np.float32)
    p = tmp_path / "obj_grasps.csv"
    GA.write_grasps_csv(str(p), T, J)
    T2, J2 = GA.read_grasps_csv(str(p), 20)
    # operator<< prints 6 significant digits
    np.testing.assert_allclose(T2, T, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(J2, J, rtol=1e-5, atol=1e-6)
    # DOF mapping permutes where the file's values land (grasp_analyzer.cpp:146-150)
    mapping = list(range(20))
    mapping[0], mapping[1] = 1, 0
    _, J3 = GA.read_grasps_csv(str(p), 20, mapping)
    np.testing.assert_allclose(J3[:, 0], J[:, 1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(J3[:, 1], J[:, 0], rtol=1e-5, atol=1e-6)


def test_pose_string_gt_and_short_lines():
    line = "7," + ",".join(str(i) for i in range(12 + 4))
    T, j, gt = GA.pose_from_string(line, 4, read_gt=True)
    assert gt == 7 and T[3] == 3 and list(j) == [12, 13, 14, 15]
    assert GA.pose_from_string("1,2,3", 4) is None            # too few values: the reference leaves the pose alone
    s = GA.string_from_pose(np.arange(12), [0.5, 0.25])
    assert s == "0,1,2,3,4,5,6,7,8,9,10,11,0.5,0.25"


def test_contact_info_and_errors(tmp_path):
    p = tmp_path / "s_o_contact_info.txt"
    p.write_text("1 1.15 0 0 0 0 1\n0 0 2.3 0 1 0 0\n\n1 0 0 -1.15 0 1 0\n")
    cp, cn, ncp, ncn = GA.read_contact_info(str(p), scale=2.0)
    np.testing.assert_allclose(cp, [[2.0, 0, 0], [0, 0, -2.0]], rtol=1e-6)
    np.testing.assert_allclose(ncp, [[0, 4.0, 0]], rtol=1e-6)
    assert cn.shape == (2, 3) and ncn.shape == (1, 3)
    e = tmp_path / "errors.csv"
    GA.write_grasp_errors(str(e), [[0.5, 2.0], [0.25, 1.0], [0.75]])
    assert e.read_text() == "0.5,2\n0.25,1\n0.75\n"
    col0 = GA.read_grasp_errors(str(e), 0, 4)
    assert list(col0[:3]) == [0.5, 0.25, 0.75] and col0[3] == np.finfo(np.float32).max
    th = tmp_path / "thumb.txt"
    th.write_text("15 7.997e-3 4.979e-3 26.407e-3")
    assert GA.read_thumb_file(str(th)) == (15, (7.997e-3, 4.979e-3, 26.407e-3))
    pm = tmp_path / "pot.txt"
    pm.write_text("0 1\n1 0\n")
    assert list(GA.read_potential_matrix(str(pm), 2)) == [0, 1, 1, 0]


def test_compare_poses():
    a = np.zeros((2, 12), dtype=np.float32)
    b = a.copy()
    b[0, 3], b[1, 11] = 1.0, 2.0
    assert GA.compare_poses(a, b) == 5.0


def test_keypoints_and_dof_sample_files(tmp_path):
    """HandFitter's inputs (hand_fitter.cpp:121-171). The reference's keypoints.txt is read where the tree is present."""
    from dart_amd import hand_fitter as HF
    p = tmp_path / "keypoints.txt"
    p.write_text("palm palm 44.7881 0 -1.27377\nindex2 index1 -22.9145 15.5859 -1.53697\n")
    kp = HF.read_keypoints(str(p), offsets={"index1": (1.0, 0.0, 0.0)})
    assert kp["palm"][0] == 0 and abs(kp["palm"][1][0][0] - 0.0447881) < 1e-9 and kp["palm"][1][1] == (0.0, 0.0, 0.0)
    assert kp["index2"][0] == 2 and abs(kp["index2"][1][0][0] + 0.0219145) < 1e-9
    d = tmp_path / "dof_samples.txt"
    d.write_text("0.1,0.2,0.3\n0.4,0.5,0.6,\n")
    np.testing.assert_allclose(HF.read_dof_samples(str(d), 3), [[0.1, 0.2, 0.3], [0.4, 0.5, 0.6]], rtol=1e-6)
    ref = "/root/reference/models/HumanHand/keypoints.txt"
    if os.path.exists(ref):
        kp = HF.read_keypoints(ref)
        assert len(kp) == 21 and kp["thumb4"][0] == 20


def test_umeyama_recovers_similarity():
    from dart_amd import hand_fitter as HF
    rng = np.random.default_rng(1)
    src = rng.normal(size=(3, 7))
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    A *= np.sign(np.linalg.det(A))
    dst = 1.3 * A @ src + np.array([[0.2], [-0.1], [0.5]])
    T = HF.umeyama(src, dst, True)
    np.testing.assert_allclose(T[:3, :3], 1.3 * A, atol=1e-9)
    np.testing.assert_allclose(T[:3, 3], [0.2, -0.1, 0.5], atol=1e-9)
    dst[:, 0] += 0.5                        # one outlier: RANSAC keeps the other six
    T2 = HF.estimate_srt(src, dst, n_iters=60, rng=np.random.default_rng(0))
    np.testing.assert_allclose(T2[:3, :3], 1.3 * A, atol=1e-6)


def test_barrett_grasp_lines_carry_the_reduced_dofs(tmp_path):
    """A reference-format *_grasps_barrett.csv / GT line holds 12 + 4 values: the Barrett pose has 4 reduced articulated dimensions
    (LinearPoseReduction(8, 4), grasp_analyzer.cpp:393-420, :427-435), and pose_from_string fills exactly those."""
    p = tmp_path / "obj_grasps_barrett.csv"
    T = np.arange(24, dtype=np.float32).reshape(2, 12)
    J = np.array([[0.1, 0.2, 0.3, 0.4], [0.5, 0.6, 0.7, 0.8]], dtype=np.float32)
    GA.write_grasps_csv(str(p), T, J)
    T2, J2 = GA.read_grasps_csv(str(p), 4, [0, 1, 2, 3])
    np.testing.assert_allclose(J2, J, rtol=1e-6)
    np.testing.assert_allclose(T2, T, rtol=1e-6)
    # an 8-DOF reader finds too few values: the reference warns and pushes the unmodified template pose (:117-133, :277-281)
    assert GA.pose_from_string(open(str(p)).readlines()[1], 8) is None
    T3, J3 = GA.read_grasps_csv(str(p), 8)
    assert T3.shape == (2, 12) and not J3.any()
    # the reduction the analyser builds for the Barrett hand
    from dart_amd import configs
    class Hand:
        joint_limits = [(0.0, 3.14), (0.0, 2.44), (0.0, 0.84), (0.0, 3.14), (0.0, 2.44), (0.0, 0.84), (0.0, 2.44), (0.0, 0.84)]
        joint_names = ["j%d" % i for i in range(8)]
    red = configs.barrett_reduction(Hand)
    full = np.zeros(8, dtype=np.float32)
    red.projectReducedToFull(np.array([1, 2, 3, 4], dtype=np.float32), full)
    assert list(full) == [1, 2, 2, 1, 3, 3, 4, 4] and red.getReducedDimensions() == 4
