"""dart::TrackerNoObs + dart::ContactPrior (src/tracker_no_obs.{h,cpp}, src/optimization/priors.h) -- host mirror.

The per-iteration work of OptimizerNoObs::optimizePoses / getError runs in libdartk (dartk_noobs_*): intersection
terms, every contact prior, the solve and the pose update stay on the device, for one pose set or for a batch of
independent candidates (ContactGrasp's grasp list, grasp_analyzer.cpp:851-902)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import _abi, device, se3
from ._abi import check, lib
from .optimizer import OptimizationOptions

F = np.float32


@dataclass
class ContactPrior:
    """ContactPrior constructor arguments (priors.h:38-49)."""
    srcModelID: int
    dstModelID: int
    srcSdfNum: int
    dstSdfNum: int          # ignored: reset to "nearest" before every evaluation (optimizer.cpp:914)
    weight: float
    contactPoint: tuple
    update_contact_point: bool = True
    update_src_pose: bool = True
    update_dst_pose: bool = True
    invert: bool = False
    contactThreshold: float = -1.0
    normal_sf: tuple = (0.0, 0.0, 0.0)

    def setWeight(self, w):
        self.weight = float(w)

    def setContactThreshold(self, t):
        self.contactThreshold = float(t)

    def isInverted(self):
        return self.invert

    def fill(self, c):
        c.srcModel, c.dstModel, c.srcSdf = int(self.srcModelID), int(self.dstModelID), int(self.srcSdfNum)
        c.weight = float(self.weight)
        for k in range(3):
            c.contactPoint[k] = float(self.contactPoint[k])
            c.normal[k] = float(self.normal_sf[k])
        c.invert = int(bool(self.invert))
        c.contactThreshold = float(self.contactThreshold)
        c.updateSrc, c.updateDst, c.updateContact = int(self.update_src_pose), int(self.update_dst_pose), int(self.update_contact_point)
        return c


@dataclass
class Point3D3DPrior:
    """Point3D3DPrior constructor arguments (priors.h:94-101): a point of a source frame is pulled to a camera-frame point."""
    srcModelID: int
    srcFrame: int
    point_c: tuple
    point_f: tuple
    weight: float

    kind = 0

    def setWeight(self, w):
        self.weight = float(w)

    def setTargetCameraPoint(self, p):
        self.point_c = tuple(float(v) for v in p)

    def getTargetCameraPoint(self):
        return self.point_c

    def getSourceFramePoint(self):
        return self.point_f

    def getSourceModelID(self):
        return self.srcModelID

    def getSourceFrame(self):
        return self.srcFrame

    def fill(self, c):
        c.kind, c.srcModel, c.srcFrame, c.weight = self.kind, int(self.srcModelID), int(self.srcFrame), float(self.weight)
        for k in range(3):
            c.point_f[k] = float(self.point_f[k])
            c.point_c[k] = float(self.point_c[k]) if k < len(self.point_c) else 0.0
        return c

    def sig(self):
        return (self.kind, self.srcModelID, self.srcFrame, self.weight, tuple(self.point_c), tuple(self.point_f))


@dataclass
class Point2D3DPrior(Point3D3DPrior):
    """Point2D3DPrior (priors.h:131-141): the point's projection is pulled to a pixel. point_c = (u, v)."""
    principalPoint: tuple = (0.0, 0.0)
    focalLength: tuple = (1.0, 1.0)

    kind = 1

    def fill(self, c):
        Point3D3DPrior.fill(self, c)
        for k in range(2):
            c.focalLength[k] = float(self.focalLength[k])
            c.principalPoint[k] = float(self.principalPoint[k])
        return c

    def sig(self):
        return Point3D3DPrior.sig(self) + (tuple(self.principalPoint), tuple(self.focalLength))


def priors_array(priors, cls=None):
    cls = cls or _abi.ContactPriorC
    arr = (cls * max(len(priors), 1))()
    for i, p in enumerate(priors):
        p.fill(arr[i])
    return arr


class TrackerNoObs:
    """TrackerNoObs (tracker_no_obs.h:23-110): models, poses, collision clouds, potential matrices, priors, options."""

    def __init__(self, max_batch=1):
        self.models: List[device.DeviceModel] = []
        self.host_models = []
        self.clouds = []
        self.potentials = []
        self.T_mc = []          # per model, camera->model (Pose::getTransformCameraToModel)
        self.joints = []        # per model: the joints, or the REDUCED articulation of a model with a pose reduction
        self.reductions = []    # per model: None | LinearPoseReduction | ParamMapPoseReduction (pose.py)
        self.damping = []       # per model: [dims, dims] damping matrix (getDampingMatrix), zero at first
        self._damping_sig = {}
        self.priors: List[ContactPrior] = []
        self.point_priors: List[Point3D3DPrior] = []      # evaluated after the contact priors (device and oracle alike)
        self.opts = OptimizationOptions()
        self.opts.lambdaIntersection = []
        self.opts.regularization = []
        self.opts.regularizationScaled = []
        self.max_batch = max_batch
        self._handle = None
        self._terms_key = None

    # ---- TrackerNoObs::addModel (tracker_no_obs.cpp:58-235): voxelised host model -> device mirror + collision cloud
    def addModel(self, host_model, collision_cloud_density=1e5, cloud_seed=0, reduction=None):
        """reduction: a pose.LinearPoseReduction / ParamMapPoseReduction (TrackerNoObs::addModel's poseReduction argument; the
        Barrett ContactGrasp set-up, grasp_analyzer.cpp:393-420): the optimisation then runs in the reduced dimensions and the
        model's `joints` entries everywhere in this class are its reduced articulation."""
        m_id = len(self.models)
        self.reductions.append(None if reduction is None or reduction.isNull() else reduction)
        dm = device.DeviceModel(host_model, model_id=m_id)
        self.models.append(dm)
        self.host_models.append(host_model)
        cloud = host_model.collision_cloud(collision_cloud_density, seed=cloud_seed)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.clouds.append(torch.from_numpy(np.ascontiguousarray(cloud, dtype=F)).to(dev))
        n = len(host_model.sdfs)
        self.potentials.append(torch.ones(n * n, dtype=torch.int32, device=dev))
        self.T_mc.append(se3.identity())
        self.joints.append(np.zeros(host_model.n_joints if self.reductions[-1] is None else self.reductions[-1].getReducedDimensions(), dtype=F))
        # _dampingMatrices.push_back(Zero(reducedDims, reducedDims)) (tracker_no_obs.cpp:214-215)
        self.damping.append(np.zeros((6 + len(self.joints[-1]),) * 2, dtype=F))
        n_models = len(self.models)
        # option vectors grow with the models (tracker_no_obs.cpp:219-229)
        old = self.opts.lambdaIntersection
        lam = [0.0] * (n_models * n_models)
        for s in range(n_models - 1):
            for d in range(n_models - 1):
                lam[s + d * n_models] = old[s + d * (n_models - 1)]
        self.opts.lambdaIntersection = lam
        self.opts.regularization.append(self.opts.regularization[0] if self.opts.regularization else 1e-20)
        self.opts.regularizationScaled.append(self.opts.regularizationScaled[0] if self.opts.regularizationScaled else 1.0)
        self._handle = None
        return m_id

    def getDampingMatrix(self, model_num):
        """TrackerNoObs::getDampingMatrix (tracker_no_obs.h:73): the [dims, dims] matrix added to the model's block of the normal
        equations (optimizer.cpp:895-896); modify it in place."""
        return self.damping[model_num]

    def getNumModels(self):
        return len(self.models)

    def getOptions(self):
        return self.opts

    def setIntersectionPotentialMatrix(self, model_num, matrix):
        n = len(self.host_models[model_num].sdfs)
        mat = np.ascontiguousarray(matrix, dtype=np.int32).reshape(-1)
        assert mat.size == n * n
        self.potentials[model_num] = torch.from_numpy(mat).to(self.clouds[model_num].device)
        self._handle = None

    def addPrior(self, prior):
        if isinstance(prior, Point3D3DPrior):
            self.point_priors.append(prior)
            self._points_sig = None
            return
        self.priors.append(prior)
        self._handle = None

    def clearPriors(self):
        """TrackerNoObs::clearPriors (tracker_no_obs.h): drops the contact and the point priors."""
        if self.priors:
            self.priors = []
            self._handle = None
        self.point_priors = []
        self._points_sig = None

    def setPose(self, model_num, T_mc, joints):
        self.T_mc[model_num] = np.ascontiguousarray(T_mc, dtype=F).reshape(12)
        self.joints[model_num] = np.ascontiguousarray(joints, dtype=F).reshape(-1)

    # ---- device state
    def _options_struct(self, cls=None):
        o = (cls or _abi.NoObsOptions)()
        n = len(self.models)
        o.numIterations, o.nModels = int(self.opts.numIterations), n
        for m in range(n):
            o.regularization[m] = self.opts.regularization[m]
            o.regularizationScaled[m] = self.opts.regularizationScaled[m]
        for i in range(n * n):
            o.lambdaIntersection[i] = self.opts.lambdaIntersection[i]
        return o

    def _ensure(self, batch):
        n = len(self.models)
        key = (tuple(v > 0 for v in self.opts.lambdaIntersection), len(self.priors), max(batch, self.max_batch),
               any(p.update_contact_point for p in self.priors))
        if self._handle is not None and key == self._terms_key:
            return
        self.close()
        self.max_batch = key[2]
        handles = (C.c_void_p * n)(*[m.handle for m in self.models])
        clouds = (C.c_void_p * n)(*[C.c_void_p(c.data_ptr()) for c in self.clouds])
        counts = (C.c_int32 * n)(*[int(c.shape[0]) for c in self.clouds])
        pots = (C.c_void_p * n)(*[C.c_void_p(p.data_ptr()) for p in self.potentials])
        lam = (C.c_float * (n * n))(*self.opts.lambdaIntersection)
        arr = priors_array(self.priors)
        h = C.c_void_p()
        reds, keep = self.reductions_struct(_abi.ReductionC)
        check(lib().dartk_noobs_create_reduced(n, handles, reds, clouds, counts, pots, lam, arr, len(self.priors), self.max_batch, C.byref(h)))
        self._handle = h
        self._damping_sig = {}
        self._terms_key = key
        self._priors_sig = self._sig()
        self._points_sig = None

    def reductions_struct(self, cls):
        """Array of dartk_reduction (or the oracle's orc_reduction, same meaning) for the models, or None when none is reduced."""
        if not any(r is not None for r in self.reductions):
            return None, []
        n = len(self.models)
        arr = (cls * n)()
        keep = []
        names = {f[0] for f in cls._fields_}
        for m, r in enumerate(self.reductions):
            if r is None:
                continue
            A = np.ascontiguousarray(r.getFirstDerivatives(), dtype=F)
            b = np.ascontiguousarray(r._b, dtype=F)
            mins = np.asarray([r.getMin(k) for k in range(r.getReducedDimensions())], dtype=F)
            maxs = np.asarray([r.getMax(k) for k in range(r.getReducedDimensions())], dtype=F)
            keep += [A, b, mins, maxs]
            mp = None
            if r.isParamMap():
                mp = np.ascontiguousarray(r.getMapping(), dtype=np.int32)
                keep.append(mp)
            if "redArtDims" in names:       # dartk_reduction
                arr[m].type, arr[m].redArtDims = (2 if r.isParamMap() else 1), r.getReducedDimensions()
                arr[m].h_A, arr[m].h_b, arr[m].h_redMin, arr[m].h_redMax = _abi.fptr(A), _abi.fptr(b), _abi.fptr(mins), _abi.fptr(maxs)
                if mp is not None:
                    arr[m].h_map = _abi.iptr(mp)
            else:                           # orc_reduction (tests)
                arr[m].type, arr[m].redArt = (2 if r.isParamMap() else 1), r.getReducedDimensions()
                arr[m].A, arr[m].b, arr[m].mins, arr[m].maxs = _abi.fptr(A), _abi.fptr(b), _abi.fptr(mins), _abi.fptr(maxs)
                if mp is not None:
                    arr[m].mapping = _abi.iptr(mp)
        return arr, keep

    def reduced_dims(self, m):
        return 6 + (self.host_models[m].n_joints if self.reductions[m] is None else self.reductions[m].getReducedDimensions())

    def _sig(self):
        return [(p.weight, tuple(p.contactPoint), p.contactThreshold, p.update_contact_point) for p in self.priors]

    def _sync_priors(self):
        for m, D in enumerate(self.damping):        # damping matrices: uploaded when they change, dropped when zero again
            sig = D.tobytes() if D.any() else None
            if self._damping_sig.get(m) != sig:
                Dc = np.ascontiguousarray(D, dtype=F)
                check(lib().dartk_noobs_set_damping(self._handle, m, Dc.ctypes.data_as(C.POINTER(C.c_float)) if sig else None, device._stream()))
                self._damping_sig[m] = sig
        psig = [p.sig() for p in self.point_priors]
        if getattr(self, "_points_sig", None) != psig and (psig or getattr(self, "_points_sig", None)):
            arr = priors_array(self.point_priors, _abi.PointPriorC)
            check(lib().dartk_noobs_set_point_priors(self._handle, arr, len(self.point_priors), device._stream()))
        self._points_sig = psig
        if self._priors_sig != self._sig():
            arr = priors_array(self.priors)
            check(lib().dartk_noobs_set_priors(self._handle, arr, len(self.priors), device._stream()))
            self._priors_sig = self._sig()

    def close(self):
        if self._handle is not None:
            lib().dartk_noobs_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- TrackerNoObs::optimizePoses (tracker_no_obs.cpp:271-299)
    def optimizePoses(self, debug=False):
        """Refines the tracker's current poses in place; returns OptimizerNoObs::getError at the result."""
        T = np.stack(self.T_mc).astype(F)[None]
        j = np.concatenate(self.joints).astype(F)[None] if sum(len(x) for x in self.joints) else np.zeros((1, 1), dtype=F)
        T_out, j_out, energy, info, first = self.optimizePosesBatch(T, j, debug=debug)
        off = 0
        for m in range(len(self.models)):
            self.T_mc[m] = T_out[0, m].copy()
            nj = len(self.joints[m])
            self.joints[m] = j_out[0, off:off + nj].copy()
            off += nj
        self.iter_info, self.first_system = (info[0] if info is not None else None), (first[0] if first is not None else None)
        if any(p.update_contact_point for p in self.priors):       # the reference moves the prior objects' contact points
            cp = self.getContactPoints(1)[0]
            for p, q in zip(self.priors, cp):
                if p.update_contact_point:
                    p.contactPoint = tuple(float(v) for v in q)
            self._priors_sig = None
        return float(energy[0])

    def getContactPoints(self, batch):
        """Contact points [batch, nPriors, 3] of the candidates after the last optimizePoses / optimizePosesBatch call (priors
        with update_contact_point move, ContactPrior::updatePriorParams, contact_prior.cpp:207-221)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        out = torch.empty((batch, len(self.priors), 3), dtype=torch.float32, device=dev)
        check(lib().dartk_noobs_contact_points(self._handle, batch, device._ptr(out), device._stream()))
        return out.cpu().numpy()

    def optimizePosesBatch(self, T_mcs, joints, debug=False, device_arrays=False):
        """`batch` independent candidates: T_mcs [B, nModels, 12], joints [B, sum of joints] (numpy or CUDA tensors).
        Returns (T_mcs, joints, energy[B], iter_info, first_system)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        T = torch.as_tensor(T_mcs, dtype=torch.float32, device=dev).contiguous().clone()
        j = torch.as_tensor(joints, dtype=torch.float32, device=dev).contiguous().clone()
        B = T.shape[0]
        self._ensure(B)
        self._sync_priors()
        o = self._options_struct()
        energy = torch.empty(B, dtype=torch.float32, device=dev)
        D = sum(self.reduced_dims(m) for m in range(len(self.models)))
        info = first = None
        if debug:
            info = torch.zeros((B, max(o.numIterations, 1), 8), dtype=torch.float32, device=dev)
            first = torch.zeros((B, D * D + D), dtype=torch.float32, device=dev)
        check(lib().dartk_noobs_optimize_poses(self._handle, C.byref(o), B, device._ptr(T), device._ptr(j), device._ptr(energy),
                                               device._ptr(info), device._ptr(first), device._stream()))
        if device_arrays:
            return T, j, energy, info, first
        return (T.cpu().numpy(), j.cpu().numpy(), energy.cpu().numpy(), None if info is None else info.cpu().numpy(),
                None if first is None else first.cpu().numpy())

    def getError(self, T_mcs=None, joints=None):
        """OptimizerNoObs::getError for the tracker's poses (or a batch)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        if T_mcs is None:
            T_mcs = np.stack(self.T_mc).astype(F)[None]
            joints = np.concatenate(self.joints).astype(F)[None] if sum(len(x) for x in self.joints) else np.zeros((1, 1), dtype=F)
        T = torch.as_tensor(T_mcs, dtype=torch.float32, device=dev).contiguous()
        j = torch.as_tensor(joints, dtype=torch.float32, device=dev).contiguous()
        B = T.shape[0]
        self._ensure(B)
        self._sync_priors()
        o = self._options_struct()
        energy = torch.empty(B, dtype=torch.float32, device=dev)
        check(lib().dartk_noobs_get_error(self._handle, C.byref(o), B, device._ptr(T), device._ptr(j), device._ptr(energy), device._stream()))
        e = energy.cpu().numpy()
        return float(e[0]) if B == 1 else e
