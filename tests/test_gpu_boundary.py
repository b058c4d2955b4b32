"""The drop-in boundary exercised the way the reference's host code would use it:
  * tests/cpp/abi_kat1 -- a plain g++ program that includes include/dartk.h, links libdartk.so and runs the reference's kernel
    test (test_obs_to_mod_kernels.cpp) plus the raw-pointer-table association (built by __graft_entry__.build());
  * the entry points that take the reference's raw device pointer tables (errorAndDataAssociationMultiModel, raycastPrediction,
    countSelfIntersections) against the handle-based ones: identical results."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import scenes
from dart_amd import _abi, configs
from dart_amd import device as D
from dart_amd._abi import check, lib
from dart_amd.optimizer import synthetic_observation

F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "abi_kat1")


def test_cpp_client_builds_and_reports_no_device_on_cpu():
    """CPU: the C++ client compiled and linked against the header + library (build()); without a GPU it stops at dartk_init."""
    assert os.path.exists(EXE), "run `python __graft_entry__.py` first (build() compiles tests/cpp/abi_kat1.cpp)"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_cpp_client_runs_reference_kernel_test")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_client_runs_reference_kernel_test(gpu):
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "34 of 80 valid points associated, 0 failed checks" in r.stdout


STUB_EXE = os.path.join(ROOT, "tests", "cpp", "stub_kat1")


def test_stub_layer_parameter_lists_equal_the_reference_headers():
    """integration/kernels_hip.cpp defines every `namespace dart` free function of obsToMod.h / modToObs.h / intersection.h /
    raycast.h with the reference's parameter list (types and names; cudaStream_t -> hipStream_t): the diff script reads the
    reference's headers as text. The reference checkout is absent on the GPU box: skipped there."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference checkout here")
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "integration", "check_signatures.py"), "/root/reference"], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout
    assert "25 functions identical, 0 problems" in r.stdout and "MISSING" not in r.stdout


def test_stub_client_builds_and_reports_no_device_on_cpu():
    assert os.path.exists(STUB_EXE), "run `python __graft_entry__.py` first (build() compiles tests/cpp/stub_kat1.cpp + integration/kernels_hip.cpp)"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_stub_client_runs_reference_kernel_test")
    r = subprocess.run([STUB_EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_stub_client_runs_reference_kernel_test(gpu):
    """the reference's kernel test in the reference's call syntax (dart::errorAndDataAssociation on a MirroredModel) + the other
    wrappers of the single-model path, through integration/kernels_hip.cpp"""
    r = subprocess.run([STUB_EXE], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "34 of 80 valid points associated" in r.stdout and "0 failed checks" in r.stdout


def _dev_ptr_table(ptrs, dev):
    return torch.tensor([int(p) for p in ptrs], dtype=torch.int64, device=dev)


@pytest.mark.gpu
def test_pointer_table_entry_points_match_handle_entry_points(gpu):
    sc = configs.config5(320, 180, 0)
    tr = configs.make_multi_tracker(sc)
    for m, (T, a) in enumerate(sc.true):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
        tr.updatePose(m)
    _, vert, norm = synthetic_observation(tr.models, sc.width, sc.height, sc.fl, sc.pp)
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
        tr.updatePose(m)
    L = lib()
    n = 2
    h, w = vert.shape[:2]
    thr, po, pn = [0.03, 0.03], [-0.03, -0.03], [[0, 0, 0], [0, 0, 0]]
    pts, counts, dbg = D.error_and_data_association_multi(vert, norm, tr.models, thr, po, pn, debug=True)
    # the tables Tracker::addModel builds (tracker.cpp:238-243) from the MirroredModel getters
    T_fm = [L.dartk_model_device_T_fm(m.handle) for m in tr.models]
    for m, p in zip(tr.models, T_fm):
        assert L.dartk_model_from_device_pointer(C.c_void_p(p)) == m.handle.value
    assert L.dartk_model_from_device_pointer(C.c_void_p(T_fm[0] + 4)) is None
    t_Tfm = _dev_ptr_table(T_fm, gpu)
    t_sf = _dev_ptr_table([L.dartk_model_device_sdf_frames(m.handle) for m in tr.models], gpu)
    t_sd = _dev_ptr_table([L.dartk_model_device_sdfs(m.handle) for m in tr.models], gpu)
    pts2 = [D.new_pts(h * w, gpu) for _ in range(n)]
    t_pts = _dev_ptr_table([p.data_ptr() for p in pts2], gpu)
    T_mcs = torch.from_numpy(np.stack([tr.getPose(m).getTransformCameraToModel() for m in range(n)]).astype(F)).to(gpu)
    d_thr, d_po = torch.tensor(thr, dtype=torch.float32, device=gpu), torch.tensor(po, dtype=torch.float32, device=gpu)
    d_pn = torch.zeros(6, dtype=torch.float32, device=gpu)
    d_ns = torch.tensor([m.n_sdfs for m in tr.models], dtype=torch.int32, device=gpu)
    last = torch.zeros(n, dtype=torch.int32, device=gpu)
    da = torch.empty((h, w), dtype=torch.int32, device=gpu)
    er = torch.empty((h, w), dtype=torch.float32, device=gpu)
    p_ = D._ptr
    check(L.dartk_assoc_multi_tables(p_(vert), p_(norm), w, h, n, p_(T_mcs), p_(t_Tfm), p_(t_sf), p_(t_sd), p_(d_ns), p_(d_thr), None,
                                     p_(d_po), p_(d_pn), p_(last), p_(t_pts), p_(da), p_(er), None, D._stream()))
    assert np.array_equal(last.cpu().numpy(), counts) and counts.min() > 100
    assert np.array_equal(da.cpu().numpy(), dbg["da"].cpu().numpy())
    for m in range(n):
        a = D.pts_to_numpy(pts[m], int(counts[m])); b = D.pts_to_numpy(pts2[m], int(counts[m]))
        a, b = a[np.argsort(a["index"])], b[np.argsort(b["index"])]
        assert np.array_equal(a["index"], b["index"]) and np.array_equal(a["dataAssociation"], b["dataAssociation"])
        assert np.array_equal(a["error"].view(np.int32), b["error"].view(np.int32))
    # raycastPrediction: one call per model with its raw arrays == the handle-based renderer
    ref = D.raycast_prediction(tr.models, tr.pred_fl, (F(tr.pred_width // 2), F(tr.pred_height // 2)), tr.pred_width, tr.pred_height)
    pred = torch.zeros_like(ref)
    for num, m in enumerate(tr.models):
        T = np.ascontiguousarray(tr.getPose(num).getTransformCameraToModel(), dtype=F)
        check(L.dartk_raycast_prediction_tables(float(tr.pred_fl[0]), float(tr.pred_fl[1]), float(tr.pred_width // 2), float(tr.pred_height // 2),
                                                tr.pred_width, tr.pred_height, num, _abi.fptr(T), C.c_void_p(T_fm[num]),
                                                C.c_void_p(L.dartk_model_device_T_mf(m.handle)), C.c_void_p(L.dartk_model_device_sdf_frames(m.handle)),
                                                C.c_void_p(L.dartk_model_device_sdfs(m.handle)), m.n_sdfs, p_(pred), 0.0, D._stream()))
    assert torch.equal(pred, ref) and int((ref[..., 2] != 0).sum()) > 100
    # countSelfIntersections
    m0 = tr.models[0]
    c_ref = D.count_self_intersections(tr.clouds[0], m0, tr.potentials[0])
    c = np.zeros(1, dtype=np.int32)
    check(L.dartk_count_self_intersections_tables(p_(tr.clouds[0]), tr.clouds[0].shape[0], C.c_void_p(L.dartk_model_device_T_mf(m0.handle)),
                                                  C.c_void_p(T_fm[0]), C.c_void_p(L.dartk_model_device_sdf_frames(m0.handle)),
                                                  C.c_void_p(L.dartk_model_device_sdfs(m0.handle)), m0.n_sdfs, p_(tr.potentials[0]), _abi.iptr(c)))
    assert int(c[0]) == c_ref
    # an array that belongs to no model is refused
    rc = L.dartk_count_self_intersections_tables(p_(tr.clouds[0]), 4, p_(vert), p_(vert), None, None, m0.n_sdfs, p_(tr.potentials[0]), _abi.iptr(c))
    assert rc == -1 and b"not a device array of a live dartk model" in L.dartk_last_error()


@pytest.mark.gpu
def test_assoc_multi_tables_T_mcs_is_an_argument_not_a_model_update(gpu):
    """The reference's T_mcs is a kernel argument (obsToMod.h:88-107): it steers the launch and leaves the MirroredModel alone. Here:
    the association with T_mcs != the models' poses equals the handle-based one with the models posed that way, and afterwards
    T_mc, T_cm (its inverse) and get_transforms() of both models are what they were."""
    sc = configs.config5(320, 180, 1)
    tr = configs.make_multi_tracker(sc)
    n = 2
    for m, (T, a) in enumerate(sc.true):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.getPose(m).getReducedArticulation()[:] = a
        tr.updatePose(m)
    _, vert, norm = synthetic_observation(tr.models, sc.width, sc.height, sc.fl, sc.pp)
    h, w = vert.shape[:2]
    thr, po, pn = [0.03, 0.03], [-0.03, -0.03], [[0, 0, 0], [0, 0, 0]]
    # expected: the models at the start poses, through the handle-based entry point
    for m, (T, a) in enumerate(sc.init):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.updatePose(m)
    pts, counts, dbg = D.error_and_data_association_multi(vert, norm, tr.models, thr, po, pn, debug=True)
    T_arg = torch.from_numpy(np.stack([tr.getPose(m).getTransformCameraToModel() for m in range(n)]).astype(F)).to(gpu)
    # the models go back to the true poses (same articulation: T_mcs replaces only T_mc), the start poses travel as T_mcs
    for m, (T, a) in enumerate(sc.true):
        tr.getPose(m).setTransformCameraToModel(T)
        tr.updatePose(m)
    before = [m.get_transforms() for m in tr.models]
    pred_before = D.raycast_prediction(tr.models, tr.pred_fl, (F(tr.pred_width // 2), F(tr.pred_height // 2)), tr.pred_width, tr.pred_height).clone()
    L = lib()
    p_ = D._ptr
    t_Tfm = _dev_ptr_table([L.dartk_model_device_T_fm(m.handle) for m in tr.models], gpu)
    pts2 = [D.new_pts(h * w, gpu) for _ in range(n)]
    t_pts = _dev_ptr_table([p.data_ptr() for p in pts2], gpu)
    d_thr, d_po = torch.tensor(thr, dtype=torch.float32, device=gpu), torch.tensor(po, dtype=torch.float32, device=gpu)
    d_pn = torch.zeros(6, dtype=torch.float32, device=gpu)
    last = torch.zeros(n, dtype=torch.int32, device=gpu)
    da = torch.empty((h, w), dtype=torch.int32, device=gpu)
    check(L.dartk_assoc_multi_tables(p_(vert), p_(norm), w, h, n, p_(T_arg), p_(t_Tfm), None, None, None, p_(d_thr), None,
                                     p_(d_po), p_(d_pn), p_(last), p_(t_pts), p_(da), None, None, D._stream()))
    assert np.array_equal(last.cpu().numpy(), counts) and counts.min() > 100
    assert np.array_equal(da.cpu().numpy(), dbg["da"].cpu().numpy())
    # not the association of the models' own (true) poses
    _, counts_true, dbg_true = D.error_and_data_association_multi(vert, norm, tr.models, thr, po, pn, debug=True)
    assert not np.array_equal(dbg_true["da"].cpu().numpy(), da.cpu().numpy())
    after = [m.get_transforms() for m in tr.models]
    for m in range(n):
        for x, y in zip(before[m], after[m]):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    # T_cm users (ray-cast, pairwise intersection transforms) see the same pose as before the call
    pred_after = D.raycast_prediction(tr.models, tr.pred_fl, (F(tr.pred_width // 2), F(tr.pred_height // 2)), tr.pred_width, tr.pred_height)
    assert torch.equal(pred_after, pred_before) and int((pred_before[..., 2] != 0).sum()) > 100
