"""GPU test of the ContactGrasp batch driver (dart_amd/grasp_analyzer.py, GraspAnalyser::load_object / analyze_grasps /
save_results, grasp_analyzer.cpp:248-389, 714-760, 851-928) on a synthetic data directory in the reference's file
formats: energies of the batch vs the CPU oracle run grasp by grasp, and the written errors file."""
import os

import numpy as np
import pytest

import oracle_lib as O
from dart_amd import configs, grasp_analyzer as GA, se3

F = np.float32
pytestmark = pytest.mark.gpu


def write_session(tmp, sc, scale):
    T_cm = np.stack([se3.invert(T) for T in sc.init_T_mc])
    GA.write_grasps_csv(os.path.join(tmp, "ellipsoid_grasps.csv"), T_cm, sc.init_joints)
    with open(os.path.join(tmp, "full1_use_ellipsoid-a_contact_info.txt"), "w") as f:
        for p in sc.priors[:-1]:
            q = np.asarray(p.contactPoint, dtype=np.float64) * 1.15 / scale
            f.write("%d %.9g %.9g %.9g %.9g %.9g %.9g\n" % (0 if p.invert else 1, q[0], q[1], q[2], *p.normal_sf))
    with open(os.path.join(tmp, "full1_use_ellipsoid-a_gt_hand_pose.txt"), "w") as f:
        f.write("0," + GA.string_from_pose(se3.invert(sc.gt_T_mc), sc.gt_joints))


def test_analyze_grasps_matches_oracle_and_writes_errors(gpu, tmp_path):
    sc = configs.contact_grasp_scene(6, n_attract=120, n_repulse=300)
    tmp = str(tmp_path)
    scale = 1.426976
    write_session(tmp, sc, scale)
    thumb = sc.priors[-1]
    an = GA.GraspAnalyser("ellipsoid-a", "full1_use", sc.hand, sc.obj, tmp, scale=scale, potential=sc.potential,
                          thumb=(thumb.srcSdfNum, thumb.contactPoint), max_batch=8)
    an.load_object()
    assert len(an.grasps_T_cm) == 6 and an.n_attract_points == 120 and an.n_repulse_points == 300
    assert len(an.tracker.priors) == 421 and an.gt_frame_poses is not None
    # the selection is a permutation of the file's points (libstdc++ shuffle order), weights are the GUI's undivided ones
    pts = sorted(tuple(np.round(p.contactPoint, 5)) for p in an.contact_priors)
    ref = sorted(tuple(np.round(np.asarray(p.contactPoint, dtype=F), 5)) for p in sc.priors[:-1])
    assert pts == ref
    assert an.contact_priors[0].weight == 150.0 and an.contact_priors[-1].weight == 20.0 and an.thumb_contact_prior.weight == 25.0
    an.set_max_iterations(6)
    mean = an.analyze_grasps(energy_type=-1)
    assert len(an.results) == 6 and all(len(r) == 2 for r in an.results)
    assert abs(mean - np.mean([r[0] for r in an.results])) < 1e-5
    # oracle, grasp by grasp, with the analyser's own prior list and options
    tr = an.tracker
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    o = tr._options_struct()
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    for c in (0, 2, 5):
        T_mc = se3.invert(an.grasps_T_cm[c])
        oT, oj, oe, _, _ = O.optimize_poses_noobs(oms, [T_mc, se3.identity()], [an.grasps_dofs[c], np.zeros(0, dtype=F)], o, clouds, pots,
                                                  tr.priors)
        print("grasp %d: energy gpu %.6f oracle %.6f, GT similarity %.6f" % (c, an.results[c][0], oe, an.results[c][1]))
        assert abs(an.results[c][0] - oe) < 1e-3 * max(1.0, abs(oe))
        assert np.abs(an.optimized[c, 24:24 + 20] - oj[0]).max() < 1e-3
    # the errors file: one line per grasp, the printed energies
    lines = open(an.errors_filename()).read().strip().split("\n")
    assert len(lines) == 6
    assert abs(float(lines[2].split(",")[0]) - an.results[2][0]) < 1e-4 * max(1.0, abs(an.results[2][0]))
    # sorting by the file's first column (set_order) and analysing the best two only
    an.analyze_grasps(n=2, order_by=0, energy_type=0, save=False)
    best = np.argsort([float(l.split(",")[0]) for l in lines])[:2]
    assert list(an.order[:2]) == list(best) and len(an.results) == 2


def test_barrett_analyser_runs_in_the_reduced_dimensions(gpu, tmp_path):
    """hand_name == "barrett": create_tracker builds LinearPoseReduction(8, 4) (grasp_analyzer.cpp:393-420), the grasp and GT files
    carry 4 DOFs per line (graspit2dartdofs is sized by getReducedArticulatedDimensions, :427-435) and the refinement runs
    on the 4 coupled parameters. Energies and refined articulations vs the oracle with the same reduction."""
    from test_gpu_contact_grasp_reduced import barrett_scene
    sc, tr0, priors, T, J = barrett_scene("linear")
    tmp = str(tmp_path)
    scale = 1.0
    # the analyser keeps the object at the camera origin: re-express the candidates in the object's frame
    T_cm = np.stack([se3.invert(se3.mul(T[c, 0], se3.invert(T[c, 1]))) for c in range(len(T))])
    GA.write_grasps_csv(os.path.join(tmp, "ellipsoid_grasps_barrett.csv"), T_cm, J)
    assert len(open(os.path.join(tmp, "ellipsoid_grasps_barrett.csv")).readlines()[1].split(",")) == 12 + 4
    with open(os.path.join(tmp, "full1_use_ellipsoid-a_contact_info.txt"), "w") as f:
        for p in priors:
            q = np.asarray(p.contactPoint, dtype=np.float64) * 1.15 / scale
            f.write("%d %.9g %.9g %.9g %.9g %.9g %.9g\n" % (0 if p.invert else 1, q[0], q[1], q[2], *p.normal_sf))
    with open(os.path.join(tmp, "barrett_full1_use_ellipsoid-a_gt_hand_pose.txt"), "w") as f:
        f.write("0," + GA.string_from_pose(T_cm[1], J[1]))
    an = GA.GraspAnalyser("ellipsoid-a", "full1_use", sc.hand, sc.obj, tmp, hand_name="barrett", scale=scale, potential=sc.potential,
                          thumb=(0, (0.0, 0.0, 0.0)), max_batch=4)
    assert an.n_dofs == 4 and an.graspit2dartdofs == [0, 1, 2, 3] and an.reduction.getReducedDimensions() == 4
    an.load_object()
    assert an.grasps_dofs.shape == (len(T), 4) and np.abs(an.grasps_dofs - J).max() < 1e-5
    assert an.gt_frame_poses is not None and len(an.gt_frame_poses) == len(sc.hand.forward_kinematics(np.zeros(8, dtype=F), "dh"))
    assert an.tracker.reduced_dims(0) == 10
    an.set_max_iterations(5)
    an.analyze_grasps(energy_type=-1, save=False)
    assert len(an.results) == len(T) and all(len(r) == 2 for r in an.results)
    # candidate 1 starts at the ground truth: the refinement moves it, but it stays the closest to GT of the candidates' starts
    tr = an.tracker
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    o = tr._options_struct()
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    with O.noobs_reductions(tr):
        for c in (0, 3):
            T_mc = se3.invert(an.grasps_T_cm[c])
            oT, oj, oe, _, _ = O.optimize_poses_noobs(oms, [T_mc, se3.identity()], [an.grasps_dofs[c], np.zeros(0, dtype=F)], o, clouds, pots,
                                                      tr.priors, dims=[10, 6])
            print("barrett grasp %d: energy gpu %.6f oracle %.6f, GT similarity %.6f" % (c, an.results[c][0], oe, an.results[c][1]))
            assert abs(an.results[c][0] - oe) < 1e-3 * max(1.0, abs(oe))
            assert an.optimized.shape[1] == 24 + 4 and np.abs(an.optimized[c, 24:28] - oj[0]).max() < 1e-3
