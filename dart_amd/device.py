"""Python host layer over the C ABI: device-resident models and the per-term entry points.

torch is used only for device memory and streams (tensors own the HBM buffers; their data_ptr() is what crosses
the C ABI). Names follow the reference: MirroredModel -> DeviceModel, the free functions of
optimization/kernels/*.h keep their names in snake case.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import check, lib

F = np.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def init(device=0):
    check(lib().dartk_init(int(device)))
    torch.cuda.set_device(device)


class DeviceModel:
    """MirroredModel (src/model/mirrored_model.h): SDFs, kinematic tables and the current pose, in HBM."""

    _next_id = 0

    def __init__(self, host_model, model_id=None, obs_layout=None):
        self.host = host_model
        if model_id is None:                     # static nextModelID counter, mirrored_model.cpp:14-20
            model_id = DeviceModel._next_id
            DeviceModel._next_id += 1
        self.model_id = model_id
        self.obs_layout = obs_layout
        desc, keep = _abi.fill_model_desc(_abi.ModelDesc, host_model, model_id, obs_layout)
        h = C.c_void_p()
        check(lib().dartk_model_create(C.byref(desc), C.byref(h)))
        self.handle = h
        self.n_frames, self.n_joints, self.n_sdfs = host_model.n_frames, host_model.n_joints, len(host_model.sdfs)
        self.dims = 6 + self.n_joints

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().dartk_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_pose(self, T_mc, joints):
        T = np.ascontiguousarray(T_mc, dtype=F).reshape(12)
        j = np.ascontiguousarray(joints, dtype=F).reshape(-1) if self.n_joints else np.zeros(1, dtype=F)
        check(lib().dartk_model_set_pose(self.handle, _abi.fptr(T), _abi.fptr(j), _stream()))

    def set_articulation(self, joints):
        j = np.ascontiguousarray(joints, dtype=F).reshape(-1) if self.n_joints else np.zeros(1, dtype=F)
        check(lib().dartk_model_set_articulation(self.handle, _abi.fptr(j), _stream()))

    def set_transforms(self, T_mc, T_mf, T_fm):
        a = np.ascontiguousarray(T_mc, dtype=F).reshape(12)
        b = np.ascontiguousarray(T_mf, dtype=F).reshape(-1)
        c = np.ascontiguousarray(T_fm, dtype=F).reshape(-1)
        assert b.size == 12 * self.n_frames and c.size == 12 * self.n_frames
        check(lib().dartk_model_set_transforms(self.handle, _abi.fptr(a), _abi.fptr(b), _abi.fptr(c), _stream()))

    def get_transforms(self):
        T_mc = np.zeros(12, dtype=F)
        T_mf = np.zeros((self.n_frames, 12), dtype=F)
        T_fm = np.zeros((self.n_frames, 12), dtype=F)
        joints = np.zeros(max(self.n_joints, 1), dtype=F)
        check(lib().dartk_model_get_transforms(self.handle, _abi.fptr(T_mc), _abi.fptr(T_mf), _abi.fptr(T_fm), _abi.fptr(joints)))
        return T_mc, T_mf, T_fm, joints[:self.n_joints]


def _get_T_mc(self):
    """camera -> model transform only (one 48-byte copy instead of the four arrays of get_transforms)"""
    T_mc = np.zeros(12, dtype=F)
    check(lib().dartk_model_get_transforms(self.handle, _abi.fptr(T_mc), None, None, None))
    return T_mc


DeviceModel.get_T_mc = _get_T_mc


def new_pts(n, device="cuda"):
    """Device array of n DataAssociatedPoint records (12 B each)."""
    return torch.empty((max(n, 1), 3), dtype=torch.int32, device=device)


def pts_to_numpy(pts, count):
    raw = pts[:count].cpu().numpy()
    out = np.empty(count, dtype=_abi.DA_DTYPE)
    out["index"] = raw[:, 0]
    out["dataAssociation"] = raw[:, 1]
    out["error"] = raw[:, 2].view(np.float32)
    return out


def pts_from_numpy(rec, device="cuda"):
    if len(rec) == 0:
        return new_pts(1, device)
    raw = np.empty((len(rec), 3), dtype=np.int32)
    raw[:, 0] = rec["index"]
    raw[:, 1] = rec["dataAssociation"]
    raw[:, 2] = rec["error"].astype(np.float32).view(np.int32)
    return torch.from_numpy(raw).to(device)


def error_and_data_association(vert, norm, model, dist_threshold=0.03, norm_threshold=-0.1, plane_offset=-0.03,
                               plane_normal=(0, 0, 0), pts=None, debug=False):
    """errorAndDataAssociation (obsToMod.h:75-86). vert/norm: float32 CUDA tensors [H,W,4].
    Returns (pts tensor, count, debug dict)."""
    h, w = vert.shape[0], vert.shape[1]
    if pts is None:
        pts = new_pts(h * w, vert.device)
    last = torch.zeros(1, dtype=torch.int32, device=vert.device)
    h_last = np.zeros(1, dtype=np.int32)
    dbg = {}
    if debug:
        dbg["da"] = torch.empty((h, w), dtype=torch.int32, device=vert.device)
        dbg["err"] = torch.empty((h, w), dtype=torch.float32, device=vert.device)
        dbg["norm"] = torch.empty((h, w, 4), dtype=torch.float32, device=vert.device)
    pn = np.asarray(plane_normal, dtype=F)
    check(lib().dartk_assoc_single(_ptr(vert), _ptr(norm), w, h, model.handle, dist_threshold, norm_threshold,
                                   plane_offset, _abi.fptr(pn), _ptr(pts), _ptr(last), _abi.iptr(h_last),
                                   _ptr(dbg.get("da")), _ptr(dbg.get("err")), _ptr(dbg.get("norm")), _stream()))
    return pts, int(h_last[0]), dbg


def error_and_data_association_multi(vert, norm, models, dist_thresholds, plane_offsets, plane_normals, pts_list=None,
                                     debug=False):
    """errorAndDataAssociationMultiModel (obsToMod.h:88-107). Returns (pts list, counts, debug dict)."""
    h, w = vert.shape[0], vert.shape[1]
    n = len(models)
    if pts_list is None:
        pts_list = [new_pts(h * w, vert.device) for _ in range(n)]
    last = torch.zeros(n, dtype=torch.int32, device=vert.device)
    handles = (C.c_void_p * n)(*[m.handle for m in models])
    table = (C.c_void_p * n)(*[C.c_void_p(p.data_ptr()) for p in pts_list])
    dt = np.asarray(dist_thresholds, dtype=F)
    po = np.asarray(plane_offsets, dtype=F)
    pn = np.ascontiguousarray(plane_normals, dtype=F).reshape(-1)
    dbg = {}
    if debug:
        dbg["da"] = torch.empty((h, w), dtype=torch.int32, device=vert.device)
        dbg["err"] = torch.empty((h, w), dtype=torch.float32, device=vert.device)
    check(lib().dartk_assoc_multi(_ptr(vert), _ptr(norm), w, h, n, handles, _abi.fptr(dt), _abi.fptr(po), _abi.fptr(pn),
                                  table, _ptr(last), _ptr(dbg.get("da")), _ptr(dbg.get("err")), _stream()))
    counts = last.cpu().numpy()
    return pts_list, counts, dbg


def sys_len(dims):
    return dims + dims * (dims + 1) // 2 + 1


def norm_eqns_obs_to_mod(vert, model, pts, n_elements, huber_delta=0.02, reduction=None, result=None, no_alloc=False):
    """normEqnsObsToMod{,Reduced,ParamMap} (obsToMod.h:12-45). reduction = None | ("linear", A tensor, redDims) |
    ("map", mapping tensor, redDims). Returns the packed system as a CUDA tensor. no_alloc: dartk_ne_obs2mod_noalloc, the
    variant that allocates nothing inside the library (then pass `result`, or the output tensor is the only allocation)."""
    h, w = vert.shape[0], vert.shape[1]
    dims = model.dims
    rd = dims if reduction is None else reduction[2]
    if result is None:
        result = torch.empty(sys_len(rd), dtype=torch.float32, device=vert.device)
    if no_alloc:
        assert reduction is None
        check(lib().dartk_ne_obs2mod_noalloc(dims, _ptr(vert), w, h, model.handle, huber_delta, _ptr(pts), n_elements, _ptr(result), _stream()))
    elif reduction is None:
        check(lib().dartk_ne_obs2mod(dims, _ptr(vert), w, h, model.handle, huber_delta, _ptr(pts), n_elements, _ptr(result), _stream()))
    elif reduction[0] == "linear":
        check(lib().dartk_ne_obs2mod_reduced(dims, rd, _ptr(reduction[1]), _ptr(vert), w, h, model.handle, huber_delta,
                                             _ptr(pts), n_elements, _ptr(result), _stream()))
    else:
        check(lib().dartk_ne_obs2mod_parammap(dims, rd, _ptr(reduction[1]), _ptr(vert), w, h, model.handle, huber_delta,
                                              _ptr(pts), n_elements, _ptr(result), _stream()))
    return result


# ---------------------------------------------------------------------------------------------- mod -> obs
def raycast_prediction(models, focal_length, principal_point, width, height, prediction=None, level_set=0.0):
    """PredictionRenderer::raytracePrediction (prediction_renderer.cpp:193-217): one launch per model, model m
    is rendered with modelNum = m (model 0 clears the map). Returns the float4 [H,W,4] prediction."""
    dev = torch.device("cuda", torch.cuda.current_device())
    if prediction is None:
        prediction = torch.zeros((height, width, 4), dtype=torch.float32, device=dev)
    for num, m in enumerate(models):
        check(lib().dartk_raycast_prediction(focal_length[0], focal_length[1], principal_point[0], principal_point[1], width,
                                             height, num, m.handle, _ptr(prediction), level_set, _stream()))
    return prediction


def cull_unobservable(prediction, obs_vert):
    ph, pw = prediction.shape[:2]
    oh, ow = obs_vert.shape[:2]
    check(lib().dartk_cull_unobservable(_ptr(prediction), pw, ph, _ptr(obs_vert), ow, oh, _stream()))
    return prediction


def generate_obs_sdf(vert, model, focal_length=537.0):
    """Optimizer::generateObsSdf (optimizer.cpp:159-219) using the model's current pose."""
    h, w = vert.shape[:2]
    check(lib().dartk_obs_sdf_build(_ptr(vert), w, h, model.handle, focal_length, _stream()))


def obs_sdf_numpy(model):
    """Host copy [D,H,W] of the model's observation SDF (MirroredModel::getObsSdf after a device->host sync)."""
    dim = model.obs_layout[0]
    out = np.empty((dim[2], dim[1], dim[0]), dtype=F)
    ptr = lib().dartk_model_obs_sdf_data(model.handle)
    check(lib().dartk_buf_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, _stream()))
    return out


def set_obs_sdf(model, data):
    """Upload observation-SDF voxel data (MirroredModel::syncObsSdfHostToDevice)."""
    dim = model.obs_layout[0]
    data = np.ascontiguousarray(data, dtype=F)
    assert data.size == dim[0] * dim[1] * dim[2]
    ptr = lib().dartk_model_obs_sdf_data(model.handle)
    check(lib().dartk_buf_h2d(C.c_void_p(ptr), data.ctypes.data_as(C.c_void_p), data.nbytes, _stream()))


def distance_transform_3d(grid_in, take_sqrt=True, signed=False):
    """distanceTransform3D / signedDistanceTransform3D (distance_transforms.h:16-29). grid_in: CUDA float tensor
    [D,H,W]; it is clobbered like the reference's input. Returns the output tensor."""
    d, h, w = grid_in.shape
    out = torch.empty_like(grid_in)
    fn = lib().dartk_signed_distance_transform_3d if signed else lib().dartk_distance_transform_3d
    check(fn(_ptr(grid_in), _ptr(out), w, h, d, int(take_sqrt), _stream()))
    return out


def norm_eqns_mod_to_obs(prediction, model, T_gc=None, truncation=None, reduction=None, result=None, counter=None):
    """normEqnsModToObs{,Truncated,Reduced,ParamMap} (modToObs.h:13-90). Returns (packed system, counter tensor)."""
    h, w = prediction.shape[:2]
    dims = model.dims
    rd = dims if reduction is None else reduction[2]
    if result is None:
        result = torch.empty(sys_len(rd), dtype=torch.float32, device=prediction.device)
    if counter is None:
        counter = torch.zeros(1, dtype=torch.int32, device=prediction.device)
    if reduction is None and truncation is None:
        T = np.ascontiguousarray(T_gc, dtype=F).reshape(12)
        check(lib().dartk_ne_mod2obs(dims, _ptr(prediction), w, h, model.handle, _abi.fptr(T), _ptr(result), _ptr(counter), _stream()))
    elif reduction is None:
        check(lib().dartk_ne_mod2obs_truncated(dims, _ptr(prediction), w, h, model.handle, truncation, _ptr(result), _ptr(counter), _stream()))
    elif reduction[0] == "linear":
        check(lib().dartk_ne_mod2obs_reduced(dims, rd, _ptr(reduction[1]), _ptr(prediction), w, h, model.handle, _ptr(result), _ptr(counter), _stream()))
    else:
        check(lib().dartk_ne_mod2obs_parammap(dims, rd, _ptr(reduction[1]), _ptr(prediction), w, h, model.handle, _ptr(result), _ptr(counter), _stream()))
    return result, counter


def error_mod_to_obs(prediction, model):
    h, w = prediction.shape[:2]
    e = np.zeros(1, dtype=F)
    check(lib().dartk_error_mod2obs(_ptr(prediction), w, h, model.handle, _abi.fptr(e), _stream()))
    return float(e[0])


# ---------------------------------------------------------------------------------------------- intersection
def norm_eqns_self_intersection(sites, model, potential, reduction=None, result=None):
    """normEqnsSelfIntersection{,Reduced,ParamMap} (intersection.h). sites: CUDA [N,4]; potential: int32 [nSdfs^2]."""
    n = sites.shape[0]
    dims = model.dims
    rd = dims if reduction is None else reduction[2]
    if result is None:
        result = torch.empty(sys_len(rd - 6), dtype=torch.float32, device=sites.device)
    if reduction is None:
        check(lib().dartk_ne_self_intersection(_ptr(sites), n, dims, model.handle, _ptr(potential), _ptr(result), _stream()))
    elif reduction[0] == "linear":
        check(lib().dartk_ne_self_intersection_reduced(_ptr(sites), n, dims, rd, model.handle, _ptr(reduction[1]), _ptr(potential), _ptr(result), _stream()))
    else:
        check(lib().dartk_ne_self_intersection_parammap(_ptr(sites), n, dims, rd, model.handle, _ptr(reduction[1]), _ptr(potential), _ptr(result), _stream()))
    return result


def norm_eqns_intersection(sites, src, dst, T_ds, T_sd, reduction=None, result=None):
    """normEqnsIntersection{,Reduced,ParamMap} (intersection.h)."""
    n = sites.shape[0]
    dims = src.dims
    rd = dims if reduction is None else reduction[2]
    if result is None:
        result = torch.empty(sys_len(rd), dtype=torch.float32, device=sites.device)
    a = np.ascontiguousarray(T_ds, dtype=F).reshape(12)
    b = np.ascontiguousarray(T_sd, dtype=F).reshape(12)
    if reduction is None:
        check(lib().dartk_ne_intersection(_ptr(sites), n, dims, _abi.fptr(a), _abi.fptr(b), src.handle, dst.handle, _ptr(result), _stream()))
    elif reduction[0] == "linear":
        check(lib().dartk_ne_intersection_reduced(_ptr(sites), n, dims, rd, _abi.fptr(a), _abi.fptr(b), src.handle, dst.handle, _ptr(reduction[1]), _ptr(result), _stream()))
    else:
        check(lib().dartk_ne_intersection_parammap(_ptr(sites), n, dims, rd, _abi.fptr(a), _abi.fptr(b), src.handle, dst.handle, _ptr(reduction[1]), _ptr(result), _stream()))
    return result


def mod_to_obs_debug_maps(prediction, model, T_gc=None, truncation=None, da=None, err=None, norm=None):
    """The debug maps of normEqnsModToObs{,Truncated,Reduced,ParamMap} (OptimizationOptions::debugModToObsDA / Err / Norm): int32
    [H*W] labels, float [H*W] residuals, float [H*W,4] gradients, updated in place (created zero-filled when not given)."""
    h, w = prediction.shape[:2]
    dev = prediction.device
    da = torch.zeros(h * w, dtype=torch.int32, device=dev) if da is None else da
    err = torch.zeros(h * w, dtype=torch.float32, device=dev) if err is None else err
    norm = torch.zeros((h * w, 4), dtype=torch.float32, device=dev) if norm is None else norm
    T = None if T_gc is None else np.ascontiguousarray(T_gc, dtype=F).reshape(12)
    check(lib().dartk_mod2obs_debug_maps(_ptr(prediction), w, h, model.handle, None if T is None else _abi.fptr(T),
                                         0 if truncation is None else 1, 0.0 if truncation is None else float(truncation),
                                         _ptr(da), _ptr(err), _ptr(norm), _stream()))
    return da, err, norm


def debug_intersection_error(sites, model, potential, pairs=()):
    """_dDebugIntersectionError of one model's collision cloud (optimizer.cpp:646-682): NaN, then the self term's residual^2, then
    += residual^2 of every pairwise term in `pairs` = [(T_ds, dst model), ...]."""
    out = torch.empty(sites.shape[0], dtype=torch.float32, device=sites.device)
    check(lib().dartk_init_debug_intersection_error(_ptr(out), sites.shape[0], _stream()))
    check(lib().dartk_debug_self_intersection_error(_ptr(sites), sites.shape[0], model.handle, _ptr(potential), _ptr(out), _stream()))
    for T_ds, dst in pairs:
        a = np.ascontiguousarray(T_ds, dtype=F).reshape(12)
        check(lib().dartk_debug_intersection_error(_ptr(sites), sites.shape[0], _abi.fptr(a), model.handle, dst.handle, _ptr(out), _stream()))
    return out


def count_self_intersections(sites, model, potential):
    c = np.zeros(1, dtype=np.int32)
    check(lib().dartk_count_self_intersections(_ptr(sites), sites.shape[0], model.handle, _ptr(potential), _abi.iptr(c), _stream()))
    return int(c[0])


def count_intersections(sites, src, dst, T_ds):
    c = np.zeros(1, dtype=np.int32)
    a = np.ascontiguousarray(T_ds, dtype=F).reshape(12)
    check(lib().dartk_count_intersections(_ptr(sites), sites.shape[0], _abi.fptr(a), src.handle, dst.handle, _abi.iptr(c), _stream()))
    return int(c[0])
