"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (never imported by dart_amd/)."""
import ctypes as C
import os
import subprocess

import numpy as np

from dart_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
F = np.float32
fp = _abi.c_float_p
ip = _abi.c_int_p
vp = C.c_void_p


class OrcGrid(C.Structure):
    _fields_ = [("dim", C.c_int32 * 3), ("offset", C.c_float * 3), ("resolution", C.c_float), ("data", fp)]


class OrcModel(C.Structure):
    _fields_ = list(_abi.ModelDesc._fields_)   # identical field order by construction (oracle/dart_oracle.h)


class OrcContactPrior(C.Structure):
    _fields_ = [("srcModel", C.c_int32), ("dstModel", C.c_int32), ("srcSdf", C.c_int32), ("weight", C.c_float),
                ("contactPoint", C.c_float * 3), ("normal", C.c_float * 3), ("invert", C.c_int32),
                ("contactThreshold", C.c_float), ("updateSrc", C.c_int32), ("updateDst", C.c_int32),
                ("updateContact", C.c_int32)]


class OrcPointPrior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("srcModel", C.c_int32), ("srcFrame", C.c_int32), ("weight", C.c_float),
                ("point_f", C.c_float * 3), ("point_c", C.c_float * 3), ("focalLength", C.c_float * 2),
                ("principalPoint", C.c_float * 2)]


_point_keep = []


def set_point_priors(point_priors):
    """orc_noobs_set_point_priors: keypoint priors for the following NoObs calls ([] clears)."""
    arr = (OrcPointPrior * max(len(point_priors), 1))()
    for k, p in enumerate(point_priors):
        p.fill(arr[k])
    _point_keep[:] = [arr]
    lib().orc_noobs_set_point_priors(arr, len(point_priors))


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_grid_value_interp.restype = C.c_float
        L.orc_error_mod2obs.restype = C.c_float
        L.orc_unpack.restype = C.c_float
        L.orc_contact_prior_contribution.restype = C.c_float
        _lib = L
    return _lib


def f(a):
    return a.ctypes.data_as(fp)


def i(a):
    return a.ctypes.data_as(ip)


def c32(a):
    return np.ascontiguousarray(a, dtype=F)


class OModel:
    """orc_model built from a dart_amd HostModel (shares the flat arrays the product uploads)."""

    def __init__(self, host_model, model_id=0, obs_layout=None):
        self.host = host_model
        self.desc, self.keep = _abi.fill_model_desc(OrcModel, host_model, model_id, obs_layout)
        self.n_frames, self.n_joints, self.n_sdfs = host_model.n_frames, host_model.n_joints, len(host_model.sdfs)
        self.dims = 6 + self.n_joints
        self.model_id = model_id

    @property
    def ref(self):
        return C.byref(self.desc)

    def fk(self, joints, mode="dh"):
        j = c32(joints) if self.n_joints else np.zeros(1, dtype=F)
        T_mf = np.zeros((self.n_frames, 12), dtype=F)
        T_fm = np.zeros((self.n_frames, 12), dtype=F)
        fn = lib().orc_fk_set_pose if mode == "dh" else lib().orc_fk_set_articulation
        fn(self.ref, f(j), f(T_mf), f(T_fm))
        return T_mf, T_fm

    def grid(self, s):
        g = self.host.sdfs[s]
        return make_grid(g.dim, g.offset, g.resolution, g.data)


def make_grid(dim, offset, res, data):
    data = np.ascontiguousarray(data, dtype=F)
    g = OrcGrid()
    for k in range(3):
        g.dim[k] = int(dim[k])
        g.offset[k] = float(offset[k])
    g.resolution = float(res)
    g.data = f(data)
    g._keep = data
    return g


def se3_invert(T):
    out = np.zeros(12, dtype=F)
    lib().orc_se3_invert(f(c32(T)), f(out))
    return out


def se3_mul(A, B):
    out = np.zeros(12, dtype=F)
    lib().orc_se3_mul(f(c32(A)), f(c32(B)), f(out))
    return out


def se3_exp(t6):
    out = np.zeros(12, dtype=F)
    lib().orc_se3_exp(f(c32(t6)), f(out))
    return out


def sys_len(d):
    return d + d * (d + 1) // 2 + 1


def assoc_single(vert, norm, om, T_mc, T_fm, dist_thr=0.03, norm_thr=-0.1, plane_offset=-0.03, plane_normal=(0, 0, 0), debug=True):
    h, w = vert.shape[:2]
    pts = np.zeros(h * w, dtype=_abi.DA_DTYPE)
    dbgDA = np.zeros((h, w), dtype=np.int32)
    dbgErr = np.zeros((h, w), dtype=F)
    dbgNorm = np.zeros((h, w, 4), dtype=F)
    pn = c32(plane_normal)
    lib().orc_assoc_single.restype = C.c_int
    n = lib().orc_assoc_single(f(c32(vert)), f(c32(norm)), w, h, om.ref, f(c32(T_mc)), f(c32(T_fm)),
                               C.c_float(dist_thr), C.c_float(norm_thr), C.c_float(plane_offset), f(pn),
                               pts.ctypes.data_as(vp), i(dbgDA), f(dbgErr), f(dbgNorm))
    return pts[:n].copy(), {"da": dbgDA, "err": dbgErr, "norm": dbgNorm}


def assoc_multi(vert, oms, T_mcs, T_fms, dist_thrs, plane_offsets, plane_normals):
    h, w = vert.shape[:2]
    n = len(oms)
    pts = [np.zeros(h * w, dtype=_abi.DA_DTYPE) for _ in range(n)]
    counts = np.zeros(n, dtype=np.int32)
    dbgDA = np.zeros((h, w), dtype=np.int32)
    dbgErr = np.zeros((h, w), dtype=F)
    models = (C.POINTER(OrcModel) * n)(*[C.pointer(o.desc) for o in oms])
    T_mcs = c32(np.asarray(T_mcs).reshape(n, 12))
    T_fm_keep = [c32(t) for t in T_fms]
    T_fm_ptrs = (fp * n)(*[f(t) for t in T_fm_keep])
    pts_ptrs = (vp * n)(*[p.ctypes.data_as(vp) for p in pts])
    lib().orc_assoc_multi(f(c32(vert)), None, w, h, n, models, f(T_mcs), T_fm_ptrs, f(c32(dist_thrs)), f(c32(plane_offsets)),
                          f(c32(np.asarray(plane_normals).reshape(-1))), pts_ptrs, i(counts), i(dbgDA), f(dbgErr))
    return [p[:c].copy() for p, c in zip(pts, counts)], counts, {"da": dbgDA, "err": dbgErr}


def _red_args(reduction):
    if reduction is None:
        return 0, 0, None, None, ()
    kind, arr, rd = reduction
    if kind == "linear":
        a = c32(arr)
        return 1, rd, f(a), None, (a,)
    m = np.ascontiguousarray(arr, dtype=np.int32)
    return 2, rd, None, i(m), (m,)


def ne_obs2mod(vert, om, T_mc, T_fm, T_mf, pts, huber_delta=0.02, reduction=None):
    red, rd, A, mp, keep = _red_args(reduction)
    d = rd if red else om.dims
    res = np.zeros(sys_len(d), dtype=F)
    pts = np.ascontiguousarray(pts)
    lib().orc_ne_obs2mod(f(c32(vert)), om.ref, f(c32(T_mc)), f(c32(T_fm)), f(c32(T_mf)), pts.ctypes.data_as(vp), len(pts),
                         C.c_float(huber_delta), red, rd, A, mp, f(res))
    return res


def error_jacobian(om, T_fm, T_mf, point_m, frame, grad_m, articulation_only=False):
    n = om.n_joints if articulation_only else om.dims
    J = np.zeros(max(n, 1), dtype=F)
    p4 = c32(list(point_m[:3]) + [1.0 if len(point_m) < 4 else point_m[3]])
    lib().orc_error_jacobian(om.ref, f(c32(T_fm)), f(c32(T_mf)), f(p4), int(frame), f(c32(grad_m)), int(articulation_only), f(J))
    return J[:n]


def model_jacobian_3d(om, T_fm, T_mf, point_m, frame):
    J = np.zeros((om.dims, 3), dtype=F)
    p4 = c32(list(point_m[:3]) + [1.0])
    lib().orc_model_jacobian_3d(om.ref, f(c32(T_fm)), f(c32(T_mf)), f(p4), int(frame), f(J))
    return J


# ---------------------------------------------------------------------------------------------- mod -> obs
def raycast_prediction(oms, poses, fl, pp, width, height, level_set=0.0):
    """poses: list of (T_mc, T_fm, T_mf) per model; returns [H,W,4]."""
    pred = np.zeros((height, width, 4), dtype=F)
    for num, (om, (T_mc, T_fm, T_mf)) in enumerate(zip(oms, poses)):
        lib().orc_raycast_prediction(C.c_float(fl[0]), C.c_float(fl[1]), C.c_float(pp[0]), C.c_float(pp[1]), width, height, num,
                                     om.ref, f(c32(T_mc)), f(c32(T_fm)), f(c32(T_mf)), C.c_float(level_set), f(pred))
    return pred


def cull_unobservable(pred, obs):
    pred = c32(pred).copy()
    lib().orc_cull_unobservable(f(pred), pred.shape[1], pred.shape[0], f(c32(obs)), obs.shape[1], obs.shape[0])
    return pred


def depth_to_vertices(depth, pp, fl, depth_range=(0.1, 5.0), out=None):
    """orc_depth_to_vertices; `out` ([h, w, 4], updated in place and returned) carries what the vertex map held before."""
    h, w = depth.shape
    vert = np.zeros((h, w, 4), dtype=F) if out is None else out
    assert vert.dtype == F and vert.flags.c_contiguous
    lib().orc_depth_to_vertices(f(c32(depth)), f(vert), w, h, C.c_float(pp[0]), C.c_float(pp[1]), C.c_float(fl[0]), C.c_float(fl[1]),
                                C.c_float(depth_range[0]), C.c_float(depth_range[1]))
    return vert


def vertices_to_normals(vert):
    h, w = vert.shape[:2]
    norm = np.zeros((h, w, 4), dtype=F)
    lib().orc_vertices_to_normals(f(c32(vert)), f(norm), w, h)
    return norm


def splat_obs_sdf(vert, T_cm, obs_layout, focal_length=537.0):
    dim, off, res = obs_layout
    data = np.zeros((dim[2], dim[1], dim[0]), dtype=F)
    g = make_grid(dim, off, res, data)
    lib().orc_splat_obs_sdf(f(c32(vert)), vert.shape[1], vert.shape[0], f(c32(T_cm)), C.byref(g), C.c_float(focal_length), f(data))
    return data


def distance_transform_3d(grid, take_sqrt=True, signed=False):
    a = c32(grid).copy()
    out = np.zeros_like(a)
    d, h, w = a.shape
    fn = lib().orc_signed_distance_transform_3d if signed else lib().orc_distance_transform_3d
    fn(f(a), f(out), w, h, d, int(take_sqrt))
    return out


def ne_mod2obs(pred, om, T_gc, T_fm, T_mf, obs_grid, truncation=None, reduction=None):
    red, rd, A, mp, keep = _red_args(reduction)
    d = rd if red else om.dims
    res = np.zeros(sys_len(d), dtype=F)
    lib().orc_ne_mod2obs.restype = C.c_int
    n = lib().orc_ne_mod2obs(f(c32(pred)), pred.shape[1], pred.shape[0], om.model_id, om.ref, f(c32(T_gc)), f(c32(T_fm)), f(c32(T_mf)),
                             C.byref(obs_grid), C.c_float(-1.0 if truncation is None else truncation), red, rd, A, mp, f(res))
    return res, n


def mod2obs_debug_maps(pred, model_num, T_gc, obs_grid, truncation=None, da=None, err=None, norm=None):
    """orc_mod2obs_debug_maps: updates (and returns) the three debug maps, created with the reference's buffers' zero fill."""
    h, w = pred.shape[:2]
    da = np.zeros(h * w, dtype=np.int32) if da is None else da
    err = np.zeros(h * w, dtype=F) if err is None else err
    norm = np.zeros((h * w, 4), dtype=F) if norm is None else norm
    lib().orc_mod2obs_debug_maps(f(c32(pred)), w, h, int(model_num), f(c32(T_gc)), C.byref(obs_grid),
                                 C.c_float(-1.0 if truncation is None else truncation), i(da), f(err), f(norm))
    return da, err, norm


def debug_intersection_error(sites_self, om, T_fm, T_mf, pot, pairs=()):
    """initDebugIntersectionError, then the self term's and every (T_ds, dst, T_fm_dst) pairwise term's debugError writes"""
    out = np.zeros(len(sites_self), dtype=F)
    lib().orc_init_debug_intersection_error(f(out), len(out))
    pot = np.ascontiguousarray(pot, dtype=np.int32)
    lib().orc_debug_self_intersection_error(f(c32(sites_self)), len(out), om.ref, f(c32(T_fm)), f(c32(T_mf)), i(pot), f(out))
    for T_ds, dst, T_fm_d in pairs:
        lib().orc_debug_intersection_error(f(c32(sites_self)), len(out), f(c32(T_ds)), om.ref, f(c32(T_mf)), dst.ref, f(c32(T_fm_d)), f(out))
    return out


def error_mod2obs(pred, obs_grid):
    return float(lib().orc_error_mod2obs(f(c32(pred)), pred.shape[1], pred.shape[0], C.byref(obs_grid)))


# ---------------------------------------------------------------------------------------------- intersection
def ne_self_intersection(sites, om, T_fm, T_mf, pot, reduction=None):
    red, rd, A, mp, keep = _red_args(reduction)
    d = (rd if red else om.dims) - 6
    res = np.zeros(sys_len(d), dtype=F)
    pot = np.ascontiguousarray(pot, dtype=np.int32)
    lib().orc_ne_self_intersection.restype = C.c_int
    n = lib().orc_ne_self_intersection(f(c32(sites)), len(sites), om.ref, f(c32(T_fm)), f(c32(T_mf)), i(pot), red, rd, A, mp, f(res))
    return res, n


def ne_intersection(sites, T_ds, T_sd, src, T_fm_s, T_mf_s, dst, T_fm_d, T_mf_d, reduction=None):
    red, rd, A, mp, keep = _red_args(reduction)
    d = rd if red else src.dims
    res = np.zeros(sys_len(d), dtype=F)
    lib().orc_ne_intersection.restype = C.c_int
    n = lib().orc_ne_intersection(f(c32(sites)), len(sites), f(c32(T_ds)), f(c32(T_sd)), src.ref, f(c32(T_fm_s)), f(c32(T_mf_s)),
                                  dst.ref, f(c32(T_fm_d)), f(c32(T_mf_d)), red, rd, A, mp, f(res))
    return res, n


def count_self_intersections(sites, om, T_fm, T_mf, pot):
    pot = np.ascontiguousarray(pot, dtype=np.int32)
    return lib().orc_count_self_intersections(f(c32(sites)), len(sites), om.ref, f(c32(T_fm)), f(c32(T_mf)), i(pot))


def count_intersections(sites, T_ds, src, T_mf_s, dst, T_fm_d):
    return lib().orc_count_intersections(f(c32(sites)), len(sites), f(c32(T_ds)), src.ref, f(c32(T_mf_s)), dst.ref, f(c32(T_fm_d)))


# ---------------------------------------------------------------------------------------------- optimizer loops
def optimize_pose(om, T_mc, joints, vert, norm, opts_struct, cloud=None, potential=None):
    """orc_optimize_pose. opts_struct: dart_amd._abi.OptOptions (same layout as orc_opt_options).
    Returns (T_mc, joints, iter_info [n,8], first_system)."""
    T = c32(T_mc).copy()
    j = c32(joints).copy() if om.n_joints else np.zeros(1, dtype=F)
    n_obs = int(np.prod(om.desc.obs_sdf_dim[:])) if om.desc.obs_sdf_dim[0] else 1
    obs = np.zeros(max(n_obs, 1), dtype=F)
    info = np.zeros((max(opts_struct.numIterations, 1), 8), dtype=F)
    first = np.zeros(3 * sys_len(om.dims), dtype=F)
    n_cloud = 0 if cloud is None else len(cloud)
    cl = c32(cloud) if n_cloud else np.zeros(4, dtype=F)
    pot = np.ascontiguousarray(potential, dtype=np.int32) if potential is not None else np.zeros(1, dtype=np.int32)
    h, w = vert.shape[:2]
    lib().orc_optimize_pose(om.ref, f(T), f(j), f(c32(vert)), f(c32(norm)), w, h, C.byref(opts_struct), f(cl), n_cloud, i(pot),
                            f(obs), f(info), f(first))
    return T, j[:om.n_joints], info, first


# ---------------------------------------------------------------------------------------------- OptimizerNoObs
class OrcNoObsOptions(C.Structure):
    _fields_ = list(_abi.NoObsOptions._fields_)


def _noobs_args(oms, T_mcs, joints, clouds, potentials, priors):
    n = len(oms)
    models = (C.POINTER(OrcModel) * n)(*[C.pointer(o.desc) for o in oms])
    T_keep = [c32(T_mcs[m]).copy() for m in range(n)]
    j_keep = [c32(joints[m]).copy() if oms[m].n_joints else np.zeros(1, dtype=F) for m in range(n)]
    T_ptrs = (fp * n)(*[f(t) for t in T_keep])
    j_ptrs = (fp * n)(*[f(j) for j in j_keep])
    c_keep = [c32(c) if len(c) else np.zeros((1, 4), dtype=F) for c in clouds]
    c_ptrs = (fp * n)(*[f(c) for c in c_keep])
    counts = (C.c_int32 * n)(*[len(c) for c in clouds])
    p_keep = [np.ascontiguousarray(p, dtype=np.int32) for p in potentials]
    p_ptrs = (ip * n)(*[i(p) for p in p_keep])
    from dart_amd.tracker_no_obs import priors_array
    arr = priors_array(priors, OrcContactPrior)
    return models, T_keep, j_keep, T_ptrs, j_ptrs, c_ptrs, counts, p_ptrs, arr, (c_keep, p_keep)


def optimize_poses_noobs(oms, T_mcs, joints, opts_struct, clouds, potentials, priors, dims=None):
    """orc_optimize_poses_noobs. opts_struct: NoObsOptions layout. Returns (T_mcs list, joints list, energy, iter_info, first).
    dims: per-model (reduced) dimensions when pose reductions are active (noobs_reductions)."""
    models, T_keep, j_keep, T_ptrs, j_ptrs, c_ptrs, counts, p_ptrs, arr, keep = _noobs_args(oms, T_mcs, joints, clouds, potentials, priors)
    D = sum(o.dims for o in oms) if dims is None else sum(dims)
    info = np.zeros((max(opts_struct.numIterations, 1), 8), dtype=F)
    first = np.zeros(D * D + D, dtype=F)
    o = OrcNoObsOptions.from_buffer_copy(bytes(opts_struct))
    lib().orc_optimize_poses_noobs.restype = C.c_float
    e = lib().orc_optimize_poses_noobs(models, T_ptrs, j_ptrs, C.byref(o), c_ptrs, counts, p_ptrs, arr, len(priors), f(info), f(first))
    return T_keep, [j_keep[m][:(oms[m].n_joints if dims is None else dims[m] - 6)] for m in range(len(oms))], float(e), info, first


def get_error_noobs(oms, T_mcs, joints, opts_struct, clouds, potentials, priors):
    models, T_keep, j_keep, T_ptrs, j_ptrs, c_ptrs, counts, p_ptrs, arr, keep = _noobs_args(oms, T_mcs, joints, clouds, potentials, priors)
    o = OrcNoObsOptions.from_buffer_copy(bytes(opts_struct))
    lib().orc_get_error_noobs.restype = C.c_float
    return float(lib().orc_get_error_noobs(models, T_ptrs, j_ptrs, C.byref(o), c_ptrs, counts, p_ptrs, arr, len(priors)))


def contact_prior_contribution(oms, prior, T_mcs, T_cms, T_fms, T_mfs, offs, dims, prior_off, JTJ, JTe):
    """orc_contact_prior_contribution: adds one prior into the dense row-major JTJ [sys, sys] / JTe [sys] (float32, in place) with
    its three contact-point unknowns at prior_off; returns (_error, destination SDF)."""
    from dart_amd.tracker_no_obs import priors_array
    n = len(oms)
    arr = priors_array([prior], OrcContactPrior)
    models = (C.POINTER(OrcModel) * n)(*[C.pointer(o.desc) for o in oms])
    keep = [[c32(x[m]) for m in range(n)] for x in (T_mcs, T_cms, T_fms, T_mfs)]
    ptrs = [(fp * n)(*[f(t) for t in k]) for k in keep]
    o_arr = (C.c_int32 * n)(*[int(v) for v in offs])
    d_arr = (C.c_int32 * n)(*[int(v) for v in dims])
    ds = C.c_int32(-1)
    assert JTJ.dtype == F and JTe.dtype == F and JTJ.flags.c_contiguous
    lib().orc_contact_prior_contribution.restype = C.c_float
    e = lib().orc_contact_prior_contribution(C.byref(arr[0]), models, ptrs[0], ptrs[1], ptrs[2], ptrs[3], o_arr, d_arr, int(prior_off),
                                             f(JTJ), int(JTJ.shape[0]), f(JTe), C.byref(ds))
    return float(e), int(ds.value)


class contact_out:
    """with contact_out(n_priors) as co: the oracle's optimisation loops leave the contact points of their priors in co.points
    [n_priors, 3] (priors with update_contact_point move, contact_prior.cpp:207-221)."""

    def __init__(self, n):
        self.points = np.zeros((max(n, 1), 3), dtype=F)

    def __enter__(self):
        lib().orc_set_contact_out(f(self.points))
        return self

    def __exit__(self, *exc):
        lib().orc_set_contact_out(None)
        return False


class damping_matrices:
    """with damping_matrices([D0, None, ...]): both optimisation loops of the oracle add model m's [dims, dims] matrix to its
    block (Tracker / TrackerNoObs::getDampingMatrix; optimizer.cpp:702-703, 895-896); None = zero."""

    def __init__(self, mats):
        self.keep = [None if m is None else np.ascontiguousarray(m, dtype=F) for m in mats]
        self.arr = (fp * len(self.keep))(*[fp() if m is None else f(m) for m in self.keep])

    def __enter__(self):
        lib().orc_set_damping(self.arr)
        return self

    def __exit__(self, *exc):
        lib().orc_set_damping(None)
        return False


def project_to_sdf_surface(grid, point_grid, threshold=1e-5, max_iters=100):
    """orc_project_to_sdf_surface (sdf.cpp:6-23) on grid coordinates; returns the projected point."""
    p = c32(point_grid).copy()
    lib().orc_project_to_sdf_surface(C.byref(grid), f(p), C.c_float(threshold), int(max_iters))
    return p


def contact_prior_update_params(oms, prior, update):
    """orc_contact_prior_update_params (contact_prior.cpp:207-221); returns the new contact point."""
    from dart_amd.tracker_no_obs import priors_array
    arr = priors_array([prior], OrcContactPrior)
    models = (C.POINTER(OrcModel) * len(oms))(*[C.pointer(o.desc) for o in oms])
    lib().orc_contact_prior_update_params(C.byref(arr[0]), f(c32(update)), models)
    return np.array(list(arr[0].contactPoint), dtype=F)


# ---------------------------------------------------------------------------------------------- Optimizer::optimizePoses
class OrcReduction(C.Structure):
    _fields_ = [("type", C.c_int32), ("redArt", C.c_int32), ("A", fp), ("b", fp), ("mapping", ip), ("mins", fp), ("maxs", fp)]


class noobs_reductions:
    """with noobs_reductions(tracker_no_obs): the oracle's NoObs calls use the tracker's pose reductions (the joints[m] arrays
    then carry reduced articulations, and the system has the reduced dimensions)."""

    def __init__(self, tracker):
        self.arr, self.keep = tracker.reductions_struct(OrcReduction)

    def __enter__(self):
        lib().orc_noobs_set_reductions(self.arr)
        return self

    def __exit__(self, *exc):
        lib().orc_noobs_set_reductions(None)
        return False


class OrcMultiOptions(C.Structure):
    _fields_ = [("numIterations", C.c_int32), ("nModels", C.c_int32), ("distThreshold", C.c_float * 8), ("planeOffset", C.c_float * 8),
                ("planeNormal", (C.c_float * 3) * 8), ("regularization", C.c_float * 8), ("regularizationScaled", C.c_float * 8),
                ("lambdaObsToMod", C.c_float), ("lambdaModToObs", C.c_float), ("lambdaIntersection", C.c_float * 64),
                ("huberDelta", C.c_float), ("focalLength", C.c_float), ("predWidth", C.c_int32), ("predHeight", C.c_int32),
                ("predFl", C.c_float * 2)]


def optimize_poses(tracker, vert, norm, T_mcs, arts, priors=()):
    """orc_optimize_poses on the state of a dart_amd.tracker.Tracker (models, reductions, clouds, options) and the contact
    priors `priors` (optimizer.cpp:727-732). Returns (T_mcs, reduced articulations, iter_info [it, model, 8], first_system)."""
    n = len(tracker.models)
    oms = [OModel(tracker.host_models[m], m, tracker.models[m].obs_layout) for m in range(n)]
    models = (C.POINTER(OrcModel) * n)(*[C.pointer(o.desc) for o in oms])
    o = tracker.opts
    mo = OrcMultiOptions()
    mo.numIterations, mo.nModels = int(o.numIterations), n
    for m in range(n):
        mo.distThreshold[m] = o.distThreshold[m]
        mo.planeOffset[m] = o.planeOffset[m]
        for k in range(3):
            mo.planeNormal[m][k] = o.planeNormal[m][k]
        mo.regularization[m] = o.regularization[m]
        mo.regularizationScaled[m] = o.regularizationScaled[m]
    for k in range(n * n):
        mo.lambdaIntersection[k] = o.lambdaIntersection[k]
    mo.lambdaObsToMod, mo.lambdaModToObs = o.lambdaObsToMod, o.lambdaModToObs
    mo.huberDelta, mo.focalLength = o.huberDelta, o.focalLength
    mo.predWidth, mo.predHeight = tracker.pred_width, tracker.pred_height
    mo.predFl[0], mo.predFl[1] = float(tracker.pred_fl[0]), float(tracker.pred_fl[1])
    reds = (OrcReduction * n)()
    keep = []
    for m in range(n):
        r = tracker.poses[m].getReduction()
        if r.isNull():
            reds[m].type, reds[m].redArt = 0, tracker.host_models[m].n_joints
            continue
        A, b = c32(r.getFirstDerivatives()), c32(r._b)
        mins = c32([r.getMin(k) for k in range(r.getReducedDimensions())])
        maxs = c32([r.getMax(k) for k in range(r.getReducedDimensions())])
        keep += [A, b, mins, maxs]
        reds[m].type = 2 if r.isParamMap() else 1
        reds[m].redArt = r.getReducedDimensions()
        reds[m].A, reds[m].b, reds[m].mins, reds[m].maxs = f(A), f(b), f(mins), f(maxs)
        if r.isParamMap():
            mp = np.ascontiguousarray(r.getMapping(), dtype=np.int32)
            keep.append(mp)
            reds[m].mapping = i(mp)
    T_keep = [c32(T_mcs[m]).copy() for m in range(n)]
    a_keep = [np.concatenate([c32(arts[m]), np.zeros(1, dtype=F)]) for m in range(n)]
    T_ptrs = (fp * n)(*[f(t) for t in T_keep])
    a_ptrs = (fp * n)(*[f(a) for a in a_keep])
    clouds = [c.cpu().numpy() for c in tracker.clouds]
    c_keep = [c32(c) if len(c) else np.zeros((1, 4), dtype=F) for c in clouds]
    c_ptrs = (fp * n)(*[f(c) for c in c_keep])
    counts = (C.c_int32 * n)(*[len(c) for c in clouds])
    p_keep = [np.ascontiguousarray(p.cpu().numpy(), dtype=np.int32) for p in tracker.potentials]
    p_ptrs = (ip * n)(*[i(p) for p in p_keep])
    obs_keep = [np.zeros(int(np.prod(tracker.models[m].obs_layout[0])), dtype=F) for m in range(n)]
    obs_ptrs = (fp * n)(*[f(x) for x in obs_keep])
    D = sum(tracker.poses[m].getReducedDimensions() for m in range(n))
    info = np.zeros((max(mo.numIterations, 1), n, 8), dtype=F)
    first = np.zeros(D * D + D, dtype=F)
    h, w = vert.shape[:2]
    from dart_amd.tracker_no_obs import priors_array
    parr = priors_array(list(priors), OrcContactPrior)
    # the tracker's damping matrices (Tracker::getDampingMatrix), None while zero
    mats = [D if D.any() else None for D in getattr(tracker, "damping", [None] * n)] if getattr(tracker, "damping", None) else [None] * n
    with damping_matrices(mats):
        lib().orc_optimize_poses(models, T_ptrs, a_ptrs, reds, C.byref(mo), f(c32(vert)), f(c32(norm)), w, h, c_ptrs, counts, p_ptrs,
                                 parr if len(priors) else None, len(priors), obs_ptrs, f(info), f(first))
    return T_keep, [a_keep[m][:len(arts[m])] for m in range(n)], info, first


# ---- mesh pipeline (oracle/mesh_oracle.c) -------------------------------------------------------------------------------
def splat_solid_mesh(vertices, faces, dim, offset, resolution, data):
    """orc_splat_solid_mesh: returns a copy of `data` ([z][y][x]) with 0 in the voxels the mesh fills."""
    v = c32(vertices).reshape(-1, 3)
    fc = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    d = np.asarray(dim, dtype=np.int32)
    out = c32(data).copy()
    rc = lib().orc_splat_solid_mesh(f(v), C.c_int(len(v)), i(fc), C.c_int(len(fc)), i(d), f(c32(offset)), C.c_float(float(F(resolution))), f(out))
    assert rc == 0
    return out


def sample_mesh(vertices, faces, density, seed=1, max_points=1 << 22):
    v = c32(vertices).reshape(-1, 3)
    fc = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((max_points, 3), dtype=F)
    n = lib().orc_sample_mesh(f(v), i(fc), C.c_int(len(fc)), C.c_float(float(F(density))), C.c_uint(seed), f(out), C.c_int(max_points))
    assert n >= 0
    return out[:n].copy()


def libc_rand(seed, n):
    out = np.zeros(n, dtype=np.int32)
    lib().orc_libc_rand(C.c_uint(seed), C.c_int(n), i(out))
    return out


# ---- test aids -------------------------------------------------------------------------------------------------------------
def set_threads(n):
    """Worker threads of the oracle's per-pixel loops (1 = the serial restatement, the default)."""
    lib().orc_set_threads(int(n))


class obs_sdf_start:
    """with obs_sdf_start(T_mcs): the oracle's optimize_pose(s) calls build their observation SDF(s) at these poses instead of
    at their own start poses -- lets a test resume a trajectory mid-way exactly as the original call would have continued."""

    def __init__(self, T_mcs):
        self.buf = c32(np.asarray(T_mcs).reshape(-1, 12)).copy()

    def __enter__(self):
        lib().orc_set_obs_sdf_start(f(self.buf))
        return self

    def __exit__(self, *exc):
        lib().orc_set_obs_sdf_start(None)
        return False
