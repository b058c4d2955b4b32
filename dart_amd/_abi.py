"""ctypes binding of libdartk.so (include/dartk.h). The product path has NO fallback: if the HIP library is
missing or fails to load, importing anything that needs it raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DARTK_LIB: a diagnostic variant of the library (tools/build_variant.py); the product is libdartk.so
LIB_PATH = os.environ.get("DARTK_LIB") or os.path.join(_HERE, "libdartk.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class DaPoint(C.Structure):
    _fields_ = [("index", C.c_int32), ("dataAssociation", C.c_int32), ("error", C.c_float)]


DA_DTYPE = np.dtype([("index", np.int32), ("dataAssociation", np.int32), ("error", np.float32)])


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("n_joints", C.c_int32), ("n_sdfs", C.c_int32), ("model_id", C.c_int32),
        ("frame_parent", c_int_p), ("joint_type", c_int_p), ("joint_axis", c_float_p), ("T_pf", c_float_p),
        ("dh", c_float_p), ("joint_limits", c_float_p), ("sdf_frame", c_int_p), ("dependencies", c_int_p),
        ("sdf_dim", c_int_p), ("sdf_offset", c_float_p), ("sdf_resolution", c_float_p),
        ("sdf_data", C.POINTER(c_float_p)),
        ("obs_sdf_dim", C.c_int32 * 3), ("obs_sdf_offset", C.c_float * 3), ("obs_sdf_resolution", C.c_float),
    ]


class OptOptions(C.Structure):
    """dartk_opt_options (include/dartk.h); orc_opt_options has the same layout."""
    _fields_ = [("numIterations", C.c_int32), ("distThreshold", C.c_float), ("normThreshold", C.c_float),
                ("planeOffset", C.c_float), ("planeNormal", C.c_float * 3), ("regularization", C.c_float),
                ("regularizationScaled", C.c_float), ("lambdaObsToMod", C.c_float), ("lambdaModToObs", C.c_float),
                ("lambdaIntersection", C.c_float), ("huberDelta", C.c_float), ("focalLength", C.c_float),
                ("predWidth", C.c_int32), ("predHeight", C.c_int32), ("predFl", C.c_float * 2)]


class ContactPriorC(C.Structure):
    """dartk_contact_prior (include/dartk.h); orc_contact_prior has the same layout."""
    _fields_ = [("srcModel", C.c_int32), ("dstModel", C.c_int32), ("srcSdf", C.c_int32), ("weight", C.c_float),
                ("contactPoint", C.c_float * 3), ("normal", C.c_float * 3), ("invert", C.c_int32),
                ("contactThreshold", C.c_float), ("updateSrc", C.c_int32), ("updateDst", C.c_int32),
                ("updateContact", C.c_int32)]


class PointPriorC(C.Structure):
    """dartk_point_prior (include/dartk.h); orc_point_prior has the same layout."""
    _fields_ = [("kind", C.c_int32), ("srcModel", C.c_int32), ("srcFrame", C.c_int32), ("weight", C.c_float),
                ("point_f", C.c_float * 3), ("point_c", C.c_float * 3), ("focalLength", C.c_float * 2),
                ("principalPoint", C.c_float * 2)]


class MultiModelC(C.Structure):
    """dartk_multi_model (include/dartk.h)"""
    _fields_ = [("model", C.c_void_p), ("redType", C.c_int32), ("redArtDims", C.c_int32), ("d_A", C.c_void_p), ("d_map", C.c_void_p),
                ("h_b", C.POINTER(C.c_float)), ("h_redMin", C.POINTER(C.c_float)), ("h_redMax", C.POINTER(C.c_float)),
                ("d_redArt", C.c_void_p), ("d_cloud", C.c_void_p), ("nCloud", C.c_int32), ("d_potential", C.c_void_p),
                ("d_pts", C.c_void_p), ("distThreshold", C.c_float), ("planeOffset", C.c_float), ("planeNormal", C.c_float * 3),
                ("regularization", C.c_float), ("regularizationScaled", C.c_float), ("d_damping", C.c_void_p)]


class MultiOptionsC(C.Structure):
    """dartk_multi_options (include/dartk.h)"""
    _fields_ = [("numIterations", C.c_int32), ("nModels", C.c_int32), ("lambdaObsToMod", C.c_float), ("lambdaModToObs", C.c_float),
                ("huberDelta", C.c_float), ("focalLength", C.c_float), ("predWidth", C.c_int32), ("predHeight", C.c_int32),
                ("predFl", C.c_float * 2), ("lambdaIntersection", C.c_float * 64)]


class ReductionC(C.Structure):
    """dartk_reduction (include/dartk.h)"""
    _fields_ = [("type", C.c_int32), ("redArtDims", C.c_int32), ("h_A", C.POINTER(C.c_float)), ("h_map", C.POINTER(C.c_int32)),
                ("h_b", C.POINTER(C.c_float)), ("h_redMin", C.POINTER(C.c_float)), ("h_redMax", C.POINTER(C.c_float))]


class NoObsOptions(C.Structure):
    """dartk_noobs_options (include/dartk.h); orc_noobs_options has the same layout."""
    _fields_ = [("numIterations", C.c_int32), ("nModels", C.c_int32), ("regularization", C.c_float * 8),
                ("regularizationScaled", C.c_float * 8), ("lambdaIntersection", C.c_float * 64)]


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


def fill_model_desc(desc_cls, host_model, model_id=0, obs_layout=None):
    """Build a ModelDesc-compatible ctypes struct (also used for the oracle's orc_model, same field order)
    from a HostModel. Returns (desc, keepalive)."""
    a = host_model.arrays()
    keep = dict(a)
    d = desc_cls()
    d.n_frames, d.n_joints, d.n_sdfs, d.model_id = host_model.n_frames, host_model.n_joints, len(host_model.sdfs), model_id
    d.frame_parent = iptr(a["frame_parent"])
    d.joint_type = iptr(a["joint_type"])
    d.joint_axis = fptr(a["joint_axis"])
    d.T_pf = fptr(a["T_pf"])
    d.dh = fptr(a["dh"])
    d.joint_limits = fptr(a["joint_limits"])
    d.sdf_frame = iptr(a["sdf_frame"])
    d.dependencies = iptr(a["dependencies"])
    d.sdf_dim = iptr(a["sdf_dim"])
    d.sdf_offset = fptr(a["sdf_offset"])
    d.sdf_resolution = fptr(a["sdf_resolution"])
    arr = (c_float_p * len(a["sdf_data"]))(*[fptr(x) for x in a["sdf_data"]])
    keep["_ptrs"] = arr
    d.sdf_data = C.cast(arr, C.POINTER(c_float_p))
    if obs_layout is not None:
        dim, off, res = obs_layout
        for k in range(3):
            d.obs_sdf_dim[k] = int(dim[k])
            d.obs_sdf_offset[k] = float(off[k])
        d.obs_sdf_resolution = float(res)
    return d, keep


COUNT_LIB_PATH = os.path.join(_HERE, "libdartk_count.so")
_lib = None
_libs = {}


def lib():
    """The library every dart_amd call goes through: libdartk.so (or, inside `with counting_library():`, its diagnostic twin)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


class counting_library:
    """with counting_library() as L: route all dart_amd calls through libdartk_count.so, the diagnostic build that counts the
    trilinear lookups its kernels perform (dartk_lookup_counters). Model handles, device buffers and streams are shared
    between the two libraries (same structs, same HIP runtime); each has its own workspace. Measurement only."""

    def __enter__(self):
        global _lib
        self.prev = lib()
        _lib = _load(COUNT_LIB_PATH)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def _load(path):
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s not built: run `python -m dart_amd.build` (the product path has no CPU fallback)" % os.path.basename(path))
    L = C.CDLL(path)
    L.dartk_last_error.restype = C.c_char_p
    L.dartk_model_obs_sdf_data.restype = C.c_void_p
    for g in ("T_mf", "T_fm", "T_mc", "sdf_frames", "sdfs", "frame_parents", "joint_types", "joint_axes"):
        fn = getattr(L, "dartk_model_device_" + g)
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p]
    L.dartk_model_from_device_pointer.restype, L.dartk_model_from_device_pointer.argtypes = C.c_void_p, [C.c_void_p]
    vp = C.c_void_p
    sigs = {
        "dartk_init": [C.c_int],
        "dartk_buf_alloc": [C.POINTER(vp), C.c_size_t],
        "dartk_buf_free": [vp],
        "dartk_buf_h2d": [vp, vp, C.c_size_t, vp],
        "dartk_buf_d2h": [vp, vp, C.c_size_t, vp],
        "dartk_buf_zero": [vp, C.c_size_t, vp],
        "dartk_stream_sync": [vp],
        "dartk_model_create": [C.POINTER(ModelDesc), C.POINTER(vp)],
        "dartk_model_destroy": [vp],
        "dartk_model_set_pose": [vp, c_float_p, c_float_p, vp],
        "dartk_model_set_articulation": [vp, c_float_p, vp],
        "dartk_model_set_transforms": [vp, c_float_p, c_float_p, c_float_p, vp],
        "dartk_model_get_transforms": [vp, c_float_p, c_float_p, c_float_p, c_float_p],
        "dartk_model_num_sdfs": [vp],
        "dartk_model_dims": [vp],
        "dartk_model_obs_sdf_data": [vp],
        "dartk_assoc_single": [vp, vp, C.c_int, C.c_int, vp, C.c_float, C.c_float, C.c_float, c_float_p, vp, vp, c_int_p, vp, vp, vp, vp],
        "dartk_assoc_multi": [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), c_float_p, c_float_p, c_float_p, C.POINTER(vp), vp, vp, vp, vp],
        "dartk_assoc_multi_tables": [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "dartk_raycast_prediction_tables": [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, c_float_p, vp, vp, vp, vp, C.c_int, vp, C.c_float, vp],
        "dartk_count_self_intersections_tables": [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, c_int_p],
        "dartk_ne_obs2mod": [C.c_int, vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp, vp],
        "dartk_ne_obs2mod_noalloc": [C.c_int, vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp, vp],
        "dartk_ne_obs2mod_reduced": [C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp, vp],
        "dartk_ne_obs2mod_parammap": [C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp, vp],
        "dartk_raycast_prediction": [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp],
        "dartk_cull_unobservable": [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp],
        "dartk_obs_sdf_build": [vp, C.c_int, C.c_int, vp, C.c_float, vp],
        "dartk_splat_obs_sdf_zeros": [vp, C.c_int, C.c_int, c_float_p, vp, C.c_float, vp],
        "dartk_distance_transform_3d": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp],
        "dartk_signed_distance_transform_3d": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp],
        "dartk_ne_mod2obs": [C.c_int, vp, C.c_int, C.c_int, vp, c_float_p, vp, vp, vp],
        "dartk_ne_mod2obs_truncated": [C.c_int, vp, C.c_int, C.c_int, vp, C.c_float, vp, vp, vp],
        "dartk_ne_mod2obs_reduced": [C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp],
        "dartk_ne_mod2obs_parammap": [C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp],
        "dartk_error_mod2obs": [vp, C.c_int, C.c_int, vp, c_float_p, vp],
        "dartk_mod2obs_debug_maps": [vp, C.c_int, C.c_int, vp, c_float_p, C.c_int, C.c_float, vp, vp, vp, vp],
        "dartk_init_debug_intersection_error": [vp, C.c_int, vp],
        "dartk_debug_self_intersection_error": [vp, C.c_int, vp, vp, vp, vp],
        "dartk_debug_intersection_error": [vp, C.c_int, c_float_p, vp, vp, vp, vp],
        "dartk_ne_self_intersection": [vp, C.c_int, C.c_int, vp, vp, vp, vp],
        "dartk_ne_self_intersection_reduced": [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp],
        "dartk_ne_self_intersection_parammap": [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp],
        "dartk_ne_intersection": [vp, C.c_int, C.c_int, c_float_p, c_float_p, vp, vp, vp, vp],
        "dartk_ne_intersection_reduced": [vp, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, vp, vp, vp, vp, vp],
        "dartk_ne_intersection_parammap": [vp, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, vp, vp, vp, vp, vp],
        "dartk_count_self_intersections": [vp, C.c_int, vp, vp, c_int_p, vp],
        "dartk_count_intersections": [vp, C.c_int, c_float_p, vp, vp, c_int_p, vp],
        "dartk_splat_solid_mesh": [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, vp, vp],
        "dartk_depth_to_vertices": [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp],
        "dartk_vertices_to_normals": [vp, vp, C.c_int, C.c_int, vp],
        "dartk_prediction_to_depth": [vp, vp, C.c_int, C.c_int, vp],
        "dartk_noobs_create": [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), c_float_p,
                               C.POINTER(ContactPriorC), C.c_int, C.c_int, C.POINTER(vp)],
        "dartk_noobs_create_reduced": [C.c_int, C.POINTER(vp), C.POINTER(ReductionC), C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), c_float_p,
                                       C.POINTER(ContactPriorC), C.c_int, C.c_int, C.POINTER(vp)],
        "dartk_noobs_destroy": [vp],
        "dartk_noobs_set_priors": [vp, C.POINTER(ContactPriorC), C.c_int, vp],
        "dartk_noobs_set_damping": [vp, C.c_int, C.POINTER(C.c_float), vp],
        "dartk_noobs_set_point_priors": [vp, C.POINTER(PointPriorC), C.c_int, vp],
        "dartk_optimize_poses": [C.POINTER(MultiOptionsC), C.POINTER(MultiModelC), vp, vp, C.c_int, C.c_int, vp, vp, vp, vp],
        "dartk_optimize_poses_priors": [C.POINTER(MultiOptionsC), C.POINTER(MultiModelC), vp, C.POINTER(ContactPriorC), C.c_int, vp, vp, C.c_int,
                                        C.c_int, vp, vp, vp, vp],
        "dartk_optimize_poses_batch": [C.POINTER(MultiOptionsC), C.POINTER(MultiModelC), C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp],
        "dartk_noobs_optimize_poses": [vp, C.POINTER(NoObsOptions), C.c_int, vp, vp, vp, vp, vp, vp],
        "dartk_noobs_get_error": [vp, C.POINTER(NoObsOptions), C.c_int, vp, vp, vp, vp],
        "dartk_noobs_contact_points": [vp, C.c_int, vp, vp],
        "dartk_lookup_counters": [C.POINTER(C.c_uint64), C.c_int],
        "dartk_profile_begin": [],
        "dartk_profile_end": [c_float_p, c_int_p],
        "dartk_optimize_pose": [vp, vp, vp, C.c_int, C.c_int, C.POINTER(OptOptions), vp, C.c_int, vp, vp, vp, vp, vp],
        "dartk_optimize_pose_batch": [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.POINTER(OptOptions), vp, C.c_int, vp, vp, vp, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        if name not in ("dartk_model_obs_sdf_data",):
            fn.restype = C.c_int
    _libs[path] = L
    return L


def declared_symbols():
    """Every `dartk_*` function name declared in include/dartk*.h (used by the CPU symbol-export test)."""
    import re
    inc = os.path.join(_HERE, "..", "include")
    names = set()
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            names.update(re.findall(r"\b(dartk_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


class DartkError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise DartkError("dartk error %d: %s" % (rc, lib().dartk_last_error().decode()))
