"""dart::OptimizationOptions and dart::Optimizer (src/optimization/optimization.h, optimizer.h) -- host mirror.

The per-iteration work runs in libdartk's device-resident loop (one C call per optimizePose); this layer only holds
options, buffers and the reference's method names / argument meaning.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import _abi, device
from ._abi import check, lib

F = np.float32


@dataclass
class OptimizationOptions:
    """Defaults of optimization.h:41-71."""
    numIterations: int = 5
    distThreshold: List[float] = field(default_factory=lambda: [0.03])
    regularization: List[float] = field(default_factory=lambda: [1e-20])
    contactRegularization: float = 1e-20
    lambdaObsToMod: float = 1.0
    lambdaModToObs: float = 1.0
    lambdaIntersection: List[float] = field(default_factory=lambda: [0.0])
    planeOffset: List[float] = field(default_factory=lambda: [-0.03])
    planeNormal: List[tuple] = field(default_factory=lambda: [(0.0, 0.0, 0.0)])
    regularizationScaled: List[float] = field(default_factory=lambda: [1.0])
    regularizationScaledThresh: float = 100.0
    contactRegularizationScaled: float = 1.0
    contactRegularizationScaledThresh: float = 100.0
    focalLength: float = 537.0
    normThreshold: float = -0.1
    huberDelta: float = 0.02
    contactThreshold: float = 0.01

    def c_struct(self, model_index, pred_w, pred_h, pred_fl, n_models=1, cls=None):
        o = (cls or _abi.OptOptions)()
        m = model_index
        o.numIterations = int(self.numIterations)
        o.distThreshold = self.distThreshold[m]
        o.normThreshold = self.normThreshold
        o.planeOffset = self.planeOffset[m]
        for k in range(3):
            o.planeNormal[k] = self.planeNormal[m][k]
        o.regularization = self.regularization[m]
        o.regularizationScaled = self.regularizationScaled[m]
        o.lambdaObsToMod = self.lambdaObsToMod
        o.lambdaModToObs = self.lambdaModToObs
        o.lambdaIntersection = self.lambdaIntersection[m + m * n_models]
        o.huberDelta = self.huberDelta
        o.focalLength = self.focalLength
        o.predWidth, o.predHeight = int(pred_w), int(pred_h)
        o.predFl[0], o.predFl[1] = float(pred_fl[0]), float(pred_fl[1])
        return o


class Optimizer:
    """dart::Optimizer (optimizer.h:74-270): owns the prediction buffer (PredictionRenderer) and debug outputs."""

    def __init__(self, depth_width, depth_height, focal_length, prediction_width=-1, prediction_height=-1):
        pw = depth_width if prediction_width == -1 else prediction_width
        ph = depth_height if prediction_height == -1 else prediction_height
        self.depth_width, self.depth_height = depth_width, depth_height
        self.pred_width, self.pred_height = pw, ph
        fl = np.asarray(focal_length, dtype=F).reshape(-1)
        if fl.size == 1:
            fl = np.array([fl[0], fl[0]], dtype=F)
        self.pred_fl = (F(F(fl[0] * F(pw)) / F(depth_width)), F(F(fl[1] * F(pw)) / F(depth_width)))   # optimizer.cpp:78
        dev = torch.device("cuda", torch.cuda.current_device())
        self.prediction = torch.zeros((ph, pw, 4), dtype=torch.float32, device=dev)
        self.iter_info = None
        self.first_system = None

    def getDevicePrediction(self):
        return self.prediction

    def optimizePose(self, model, vert, norm, opts, collision_cloud=None, potential=None, debug=False):
        """Optimizer::optimizePose (optimizer.cpp:423-522). The model's current device pose is the start and
        receives the result (read it with model.get_transforms())."""
        h, w = vert.shape[:2]
        o = opts.c_struct(0, self.pred_width, self.pred_height, self.pred_fl)
        n_cloud = 0 if collision_cloud is None else collision_cloud.shape[0]
        info = first = None
        if debug:
            info = torch.zeros((max(opts.numIterations, 1), 8), dtype=torch.float32, device=vert.device)
            first = torch.zeros(3 * device.sys_len(model.dims), dtype=torch.float32, device=vert.device)
        check(lib().dartk_optimize_pose(model.handle, device._ptr(vert), device._ptr(norm), w, h, C.byref(o),
                                        device._ptr(collision_cloud), n_cloud, device._ptr(potential),
                                        device._ptr(self.prediction), device._ptr(info), device._ptr(first), device._stream()))
        self.iter_info, self.first_system = info, first

    def optimizePoseBatch(self, model, vert, norm, T_mcs, joints, opts, collision_cloud=None, potential=None, debug=False):
        """Optimizer::optimizePose (optimizer.cpp:423-522) for independent frames side by side (the reference loops over the frames
        on the host): vert / norm [nFrames, h, w, 4], T_mcs [nFrames, 12] and joints [nFrames, nJoints] device tensors, updated in
        place. Frame f's result equals optimizePose on a model set to (T_mcs[f], joints[f]). The model's own pose is untouched."""
        n, h, w = vert.shape[:3]
        assert vert.is_contiguous() and norm.is_contiguous() and norm.shape == vert.shape
        assert T_mcs.shape == (n, 12) and T_mcs.dtype == torch.float32 and T_mcs.is_contiguous()
        nj = model.dims - 6
        assert nj == 0 or (joints.shape == (n, nj) and joints.dtype == torch.float32 and joints.is_contiguous())
        o = opts.c_struct(0, self.pred_width, self.pred_height, self.pred_fl)
        n_cloud = 0 if collision_cloud is None else collision_cloud.shape[0]
        info = torch.zeros((n, max(opts.numIterations, 1), 8), dtype=torch.float32, device=vert.device) if debug else None
        check(lib().dartk_optimize_pose_batch(model.handle, n, device._ptr(vert), device._ptr(norm), w, h, C.byref(o),
                                              device._ptr(collision_cloud), n_cloud, device._ptr(potential), device._ptr(T_mcs),
                                              device._ptr(joints) if nj else None, device._ptr(info), device._stream()))
        self.iter_info = info


def depth_to_vertices(depth, pp, fl, depth_range=(0.1, 5.0), out=None):
    """PointCloudSource::projectDepthMap -> depthToVertices (point_cloud_src.h:261-285). `out`: an existing [h, w, 4] vertex
    map to update; pixels whose depth is out of range only get w = 0 and keep their x, y, z (organized_point_cloud.cu:36-38)."""
    h, w = depth.shape
    vert = torch.zeros((h, w, 4), dtype=torch.float32, device=depth.device) if out is None else out
    assert vert.shape == (h, w, 4) and vert.dtype == torch.float32 and vert.is_contiguous()
    check(lib().dartk_depth_to_vertices(device._ptr(depth), device._ptr(vert), w, h, pp[0], pp[1], fl[0], fl[1],
                                        depth_range[0], depth_range[1], device._stream()))
    return vert


def vertices_to_normals(vert):
    h, w = vert.shape[:2]
    norm = torch.empty_like(vert)
    check(lib().dartk_vertices_to_normals(device._ptr(vert), device._ptr(norm), w, h, device._stream()))
    return norm


def synthetic_observation(models, width, height, fl, pp, noise_sigma=0.0, seed=0):
    """On-device synthetic depth (north_star: replaces the OpenGL render / sensor path): SDF ray-cast of the posed
    models -> depth image -> vertex map (back-projection) -> normal map. Returns (depth, vert, norm)."""
    pred = device.raycast_prediction(models, fl, pp, width, height)
    depth = torch.empty((height, width), dtype=torch.float32, device=pred.device)
    check(lib().dartk_prediction_to_depth(device._ptr(pred), device._ptr(depth), width, height, device._stream()))
    if noise_sigma > 0:
        g = torch.Generator(device=pred.device)
        g.manual_seed(seed)
        depth = torch.where(depth > 0, depth + noise_sigma * torch.randn(depth.shape, generator=g, device=pred.device), depth)
    vert = depth_to_vertices(depth, pp, fl)
    norm = vertices_to_normals(vert)
    return depth, vert, norm
