"""HandMapper (remap_hand.cpp): HumanHand pose -> Allegro pose through chain-point Point3D3DPriors in the NoObs optimiser,
GPU vs the oracle on the same priors, plus the pose-file formats. No reference test covers it: parity unpinned."""
import os

import numpy as np
import pytest

import oracle_lib as O
from dart_amd import configs, se3
from dart_amd import hand_mapper as HM
from dart_amd.model import load_asset
from test_gpu_optimize_pose import pose_diff

F = np.float32


def test_pose_line_files(tmp_path):
    T = se3.mul(se3.translation(0.1, -0.2, 0.3), se3.from_euler(0.3, 0.2, -0.1)).astype(F)
    dofs = np.linspace(-0.2, 0.9, 20).astype(F)
    p = str(tmp_path / "0_use_mug_gt_hand_pose.txt")
    HM.write_pose_line(p, T, dofs)
    line = open(p).read().strip().split(",")
    assert line[0] == "-1" and len(line) == 1 + 12 + 20
    T2, d2 = HM.read_pose_line(p)
    assert np.allclose(T2, T, atol=1e-6) and np.allclose(d2, dofs, atol=1e-6)
    assert HM.mapped_filename("0_use_mug_gt_hand_pose.txt") == "allegro_mapped_0_use_mug_gt_hand_pose.txt"
    assert HM.mapped_filename("graspit_0_use_mug_gt_hand_pose.txt", graspit=True) == "graspit_allegro_mapped_0_use_mug_gt_hand_pose.txt"


@pytest.mark.gpu
def test_mapping_matches_oracle_and_pulls_fingers_to_targets(gpu, tmp_path):
    src = load_asset("human_hand").voxelize(0.005, 0.10)
    dst = load_asset("allegro").voxelize(0.005, 0.10)
    pot, _ = configs._hand_potential("allegro", len(dst.sdfs))
    hm = HM.HandMapper(src, dst, pot, data_dir=str(tmp_path))
    hm.set_max_iterations(6)
    rng = np.random.default_rng(3)
    lim = np.asarray(src.joint_limits, dtype=np.float64).reshape(-1, 2)
    dofs = (lim[:, 0] + (lim[:, 1] - lim[:, 0]) * rng.uniform(0.3, 0.6, len(lim))).astype(F)
    T_cm = se3.mul(se3.translation(0.02, -0.03, 0.6), se3.from_euler(0.4, -0.2, 0.1)).astype(F)
    HM.write_pose_line(str(tmp_path / "0_use_mug_gt_hand_pose.txt"), T_cm, dofs)
    hm.set_src_pose(pose_filename="0_use_mug_gt_hand_pose.txt")
    assert np.allclose(hm.src_T_cm, T_cm, atol=1e-6)

    # the start poses and the priors, as do_mapping builds them
    tr = hm.tracker
    tr.setPose(0, se3.invert(hm.src_T_cm), hm.src_dofs)
    tr.setPose(1, se3.invert(se3.mul(hm.src_T_cm, hm.T_s_d).astype(F)), np.zeros(dst.n_joints, dtype=F))
    priors = hm.create_priors()
    frames = sorted(set(p.getSourceFrame() for p in priors))
    assert len(priors) > 100 and all((f - 1) % 4 != 3 for f in frames) and len(frames) >= 8
    T0 = [tr.T_mc[m].copy() for m in range(2)]
    j0 = [tr.joints[m].copy() for m in range(2)]

    def prior_error(T_mc, joints):
        T_cm_d = se3.invert(np.asarray(T_mc, dtype=F))
        T_mf = np.asarray(dst.forward_kinematics(np.asarray(joints, dtype=F), "dh"), dtype=F).reshape(-1, 12)
        e = [np.linalg.norm(se3.apply(se3.mul(T_cm_d, T_mf[p.getSourceFrame()]), np.asarray([p.getSourceFramePoint()], dtype=F))[0] -
                            np.asarray(p.getTargetCameraPoint(), dtype=F)) for p in priors]
        return float(np.mean(e))

    e_before = prior_error(T0[1], j0[1])
    energy = hm.do_mapping()
    e_after = prior_error(tr.T_mc[1], tr.joints[1])
    print("mean chain-point distance %.4f -> %.4f m, energy %.5f" % (e_before, e_after, energy))
    assert e_after < 0.8 * e_before          # 6 of the default 160 iterations of the triangular (A-1) solve
    # the source hand carries no term: it stays where it was put
    assert np.abs(tr.T_mc[0] - T0[0]).max() < 1e-6 and np.abs(tr.joints[0] - j0[0]).max() < 1e-6

    oms = [O.OModel(src, 0), O.OModel(dst, 1)]
    clouds = [c.cpu().numpy() for c in tr.clouds]
    pots = [p.cpu().numpy() for p in tr.potentials]
    O.set_point_priors(tr.point_priors)
    try:
        oT, oj, oe, oinfo, ofirst = O.optimize_poses_noobs(oms, T0, j0, tr._options_struct(), clouds, pots, [])
    finally:
        O.set_point_priors([])
    dt, dr = pose_diff(tr.T_mc[1], oT[1])
    dj = float(np.abs(tr.joints[1] - oj[1]).max())
    print("vs oracle after 6 iterations: |dt|=%.2e |dr|=%.2e max|dq|=%.2e; energy gpu %.6f oracle %.6f" % (dt, dr, dj, energy, oe))
    assert dt < 1e-3 and dr < 1e-3 and dj < 1e-3
    assert abs(energy - oe) < 1e-3 * max(1.0, abs(oe))
    out = hm.save_result("0_use_mug_gt_hand_pose.txt")
    assert os.path.basename(out) == "allegro_mapped_0_use_mug_gt_hand_pose.txt"
    T_saved, d_saved = HM.read_pose_line(out)
    assert np.allclose(T_saved, se3.invert(tr.T_mc[1]), atol=1e-5) and np.allclose(d_saved, tr.joints[1], atol=1e-5)
