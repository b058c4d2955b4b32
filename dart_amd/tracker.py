"""dart::Tracker + Optimizer::optimizePoses (src/tracker.{h,cpp}, src/optimization/optimizer.cpp:524-781) -- host mirror
of the multi-model tracking loop with observations and (optionally) reduced poses.

Two forms: optimizePosesDevice() hands the whole loop to libdartk (dartk_optimize_poses: terms, block assembly, the
triangular solve, reduced pose updates, projection and forward kinematics stay on the device); optimizePoses() drives the
loop from the host exactly as the reference does -- every term one C-ABI call (association, normal equations,
intersection, ray-cast, obs-SDF), one copy back per term, assembly / solve / update in fp32 numpy following optimizer.cpp
line by line. The second is the readable restatement the first is tested against."""
import numpy as np
import torch

from . import device, se3
from .optimizer import OptimizationOptions
from .pose import NullReduction, Pose

F = np.float32


def unpack(eJ, JTJ, sys_, mult, dims):
    """Optimizer::unpack (optimizer.cpp:50-70): eJ += mult*J^T e; upper JTJ(i,j) += mult*packed[j(j+1)/2+i]; returns mult*e."""
    m = F(mult)
    eJ += m * sys_[:dims]
    packed = sys_[dims:dims + dims * (dims + 1) // 2]
    for j in range(dims):
        base = (j * (j + 1)) >> 1
        JTJ[:j + 1, j] += m * packed[base:base + j + 1]
    return F(m * sys_[dims + dims * (dims + 1) // 2])


def solve_upper(U, b):
    """Eigen SparseMatrix::triangularView<Upper>().solve (optimizer.cpp:732): back substitution, columns last to first, a
    column is skipped when its rhs entry is exactly 0 (SURVEY A-1)."""
    n = len(b)
    x = b.astype(F).copy()
    for i in range(n - 1, -1, -1):
        t = x[i]
        if t != 0:
            x[i] = F(t / U[i, i])
            t = x[i]
            col = U[:i, i]
            nz = col != 0
            x[:i][nz] = (x[:i][nz] - t * col[nz]).astype(F)
    return x


class Tracker:
    """Tracker (tracker.h:25-130): models, poses (with reductions), collision clouds, potential matrices, options."""

    def __init__(self, depth_width, depth_height, focal_length, pred_width=None, pred_height=None):
        self.width, self.height = depth_width, depth_height
        self.fl = (F(focal_length), F(focal_length)) if np.isscalar(focal_length) else tuple(F(v) for v in focal_length)
        self.pred_width = depth_width // 2 if pred_width is None else pred_width       # tracker.cpp:280
        self.pred_height = depth_height // 2 if pred_height is None else pred_height
        self.pred_fl = (F(F(self.fl[0] * F(self.pred_width)) / F(depth_width)), F(F(self.fl[1] * F(self.pred_width)) / F(depth_width)))
        self.models, self.host_models, self.poses, self.clouds, self.potentials = [], [], [], [], []
        self.red_dev = []     # per model: None | ("linear", A tensor, redDims) | ("map", mapping tensor, redDims)
        self.opts = OptimizationOptions()
        self.opts.lambdaIntersection = []
        self.opts.regularization, self.opts.regularizationScaled = [], []
        self.opts.distThreshold, self.opts.planeOffset, self.opts.planeNormal = [], [], []
        self.priors = []
        self.damping = []     # per model: [dims, dims] damping matrix (getDampingMatrix), zero at first
        self._ra_ring = []    # per model: pinned staging slots of the reduced articulation (_upload_red_art)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.prediction = torch.zeros((self.pred_height, self.pred_width, 4), dtype=torch.float32, device=dev)
        self.iteration_summaries = None
        self.first_system = None

    def addPrior(self, prior):
        """Tracker::addPrior (tracker.h): a tracker_no_obs.ContactPrior evaluated in every iteration of optimizePoses."""
        self.priors.append(prior)

    def clearPriors(self):
        self.priors = []

    def getDampingMatrix(self, model_num):
        """Tracker::getDampingMatrix (tracker.h:122): the [dims, dims] matrix added to the model's block of the normal equations
        after the regularisation and the LM scaling (optimizer.cpp:702-703); modify it in place."""
        return self.damping[model_num]

    _RA_SLOTS = 8

    def _upload_red_art(self, m, values, dev):
        """Model m's reduced articulation as a fresh device array, uploaded through the model's ring of pinned slots without
        waiting: a copy from pageable memory would hold the host until the stream -- the previous frame's loop -- has drained, and
        the next frame's set-up could not be enqueued behind it. (The ring is made when the model is added; a slot is reused
        once the copy that last read it has run.)"""
        ring = self._ra_ring[m]
        k = ring["next"]
        ring["next"] = (k + 1) % self._RA_SLOTS
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()
        slot = ring["pin"][k, :len(values)]
        slot.copy_(torch.from_numpy(values))
        out = slot.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["ev"][k] = ev
        return out

    def _zero1(self, dev):
        """a one-float device array for models without articulation (allocated and cleared once, not per call)"""
        z = getattr(self, "_zero1_dev", None)
        if z is None or z.device != torch.device(dev):
            z = self._zero1_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        return z

    def _damping_dev(self, m, keep):
        """device pointer of model m's damping matrix, None while it is zero"""
        D = self.damping[m]
        if not D.any():
            return None
        t = torch.from_numpy(np.ascontiguousarray(D, dtype=F)).to(self.prediction.device)
        keep.append(t)
        return device._ptr(t)

    def addModel(self, host_model, reduction=None, collision_cloud_density=1e5, cloud_seed=0):
        """Tracker::addModel (tracker.cpp:59-267): device mirror incl. the 64^3 observation SDF, collision cloud, pose."""
        m_id = len(self.models)
        dev = self.prediction.device
        self.models.append(device.DeviceModel(host_model, model_id=m_id, obs_layout=host_model.obs_sdf_layout()))
        self.host_models.append(host_model)
        lim = np.asarray(host_model.joint_limits, dtype=F).reshape(-1, 2)
        red = reduction or NullReduction(host_model.n_joints, lim[:, 0], lim[:, 1], host_model.joint_names)
        self.poses.append(Pose(red))
        rd = 6 + red.getReducedDimensions()
        self.damping.append(np.zeros((rd, rd), dtype=F))      # _dampingMatrices.push_back(Zero(...)) (tracker.cpp:246-247)
        self._ra_ring.append({"pin": torch.empty((self._RA_SLOTS, max(rd - 6, 1)), dtype=torch.float32).pin_memory(),
                              "ev": [None] * self._RA_SLOTS, "next": 0})
        if red.isNull():
            self.red_dev.append(None)
        elif red.isParamMap():
            self.red_dev.append(("map", torch.from_numpy(np.ascontiguousarray(red.getMapping(), dtype=np.int32)).to(dev), rd))
        else:
            self.red_dev.append(("linear", torch.from_numpy(np.ascontiguousarray(red.getFirstDerivatives(), dtype=F)).to(dev), rd))
        cloud = host_model.collision_cloud(collision_cloud_density, seed=cloud_seed)
        self.clouds.append(torch.from_numpy(np.ascontiguousarray(cloud, dtype=F)).to(dev))
        n_s = len(host_model.sdfs)
        self.potentials.append(torch.ones(n_s * n_s, dtype=torch.int32, device=dev))
        n = len(self.models)
        old = self.opts.lambdaIntersection
        lam = [0.0] * (n * n)
        for s in range(n - 1):
            for d in range(n - 1):
                lam[s + d * n] = old[s + d * (n - 1)]
        self.opts.lambdaIntersection = lam
        self.opts.regularization.append(self.opts.regularization[0] if self.opts.regularization else 1e-20)
        self.opts.regularizationScaled.append(self.opts.regularizationScaled[0] if self.opts.regularizationScaled else 1.0)
        self.opts.distThreshold.append(self.opts.distThreshold[0] if self.opts.distThreshold else 0.03)
        self.opts.planeOffset.append(-0.03)
        self.opts.planeNormal.append((0.0, 0.0, 0.0))
        return m_id

    def getNumModels(self):
        return len(self.models)

    def getOptions(self):
        return self.opts

    def getPose(self, m):
        return self.poses[m]

    def setIntersectionPotentialMatrix(self, m, matrix):
        self.potentials[m] = torch.from_numpy(np.ascontiguousarray(matrix, dtype=np.int32).reshape(-1)).to(self.prediction.device)

    def updatePose(self, m):
        """Tracker::updatePose: projectReducedToFull + MirroredModel::setPose."""
        p = self.poses[m]
        p.projectReducedToFull()
        self.models[m].set_pose(p.getTransformCameraToModel(), p.getArticulation())

    # ---- Optimizer::optimizePoses (optimizer.cpp:524-781), device-resident (dartk_optimize_poses)
    def optimizePosesDevice(self, vert, norm, debug=False, readback=True, from_device=False):
        """The whole loop in one library call: no host round trip per term or per iteration. Poses come back at the end
        (readback=False: they stay on the device -- the models' device poses and the returned reduced articulations hold the
        result -- and the call does not wait for the stream; readPoses() fetches them when the caller needs them).
        from_device=True starts from what the previous call left on the device instead of uploading the host-side poses: a
        tracker that runs frame after frame with readback=False, from_device=True never moves a pose across the bus."""
        import ctypes as C
        from . import _abi
        from ._abi import check, lib
        o, n = self.opts, len(self.models)
        dev = self.prediction.device
        h, w = vert.shape[:2]
        mm = (_abi.MultiModelC * n)()
        keep = []
        red_art = []
        carried = getattr(self, "_red_art_dev", None) if from_device else None
        if from_device and (carried is None or len(carried) != n):
            raise ValueError("from_device=True needs the poses a previous optimizePosesDevice call left on the device")
        for m in range(n):
            p, red = self.poses[m], self.red_dev[m]
            ra_n = p.getReducedArticulatedDimensions()
            if carried is not None:
                ra = carried[m]
            else:
                self.updatePose(m)
                ra = self._upload_red_art(m, np.ascontiguousarray(p.getReducedArticulation(), dtype=F).reshape(-1)[:ra_n], dev) if ra_n \
                    else self._zero1(dev)
            red_art.append(ra)
            mins = np.asarray([p.getReducedMin(i) for i in range(ra_n)] or [0.0], dtype=F)
            maxs = np.asarray([p.getReducedMax(i) for i in range(ra_n)] or [0.0], dtype=F)
            b = np.ascontiguousarray(getattr(p._reduction, "_b", np.zeros(self.host_models[m].n_joints, dtype=F)), dtype=F)
            if getattr(self, "_pts_dev", None) is None or len(self._pts_dev) != n or self._pts_dev[m].shape[0] < w * h:
                self._pts_dev = [device.new_pts(w * h, dev) for _ in range(n)]
            keep += [mins, maxs, b]
            c = mm[m]
            c.model = self.models[m].handle
            c.redType = 0 if red is None else (2 if red[0] == "map" else 1)
            c.redArtDims = ra_n
            c.d_A = device._ptr(red[1]) if red is not None and red[0] == "linear" else None
            c.d_map = device._ptr(red[1]) if red is not None and red[0] == "map" else None
            c.h_b, c.h_redMin, c.h_redMax = _abi.fptr(b), _abi.fptr(mins), _abi.fptr(maxs)
            c.d_redArt = device._ptr(ra)
            c.d_cloud, c.nCloud = device._ptr(self.clouds[m]), int(self.clouds[m].shape[0])
            c.d_potential = device._ptr(self.potentials[m])
            c.d_pts = device._ptr(self._pts_dev[m])
            c.distThreshold, c.planeOffset = float(o.distThreshold[m]), float(o.planeOffset[m])
            for k in range(3):
                c.planeNormal[k] = float(o.planeNormal[m][k])
            c.regularization, c.regularizationScaled = float(o.regularization[m]), float(o.regularizationScaled[m])
            c.d_damping = self._damping_dev(m, keep)
        mo = _abi.MultiOptionsC()
        mo.numIterations, mo.nModels = int(o.numIterations), n
        mo.lambdaObsToMod, mo.lambdaModToObs = float(o.lambdaObsToMod), float(o.lambdaModToObs)
        mo.huberDelta, mo.focalLength = float(o.huberDelta), float(o.focalLength)
        mo.predWidth, mo.predHeight = self.pred_width, self.pred_height
        mo.predFl[0], mo.predFl[1] = float(self.pred_fl[0]), float(self.pred_fl[1])
        for i in range(n * n):
            mo.lambdaIntersection[i] = float(o.lambdaIntersection[i])
        D = int(sum(p.getReducedDimensions() for p in self.poses))
        info = first = None
        if debug:
            info = torch.zeros((max(o.numIterations, 1), n, 8), dtype=torch.float32, device=dev)
            first = torch.zeros(D * D + D, dtype=torch.float32, device=dev)
        if self.priors:         # Tracker::addPrior: contact priors evaluated and updated inside the loop (optimizer.cpp:727-732,757-759)
            from .tracker_no_obs import priors_array
            arr = priors_array(self.priors)
            d_pri = torch.empty(C.sizeof(_abi.ContactPriorC) * len(self.priors), dtype=torch.uint8, device=dev)
            check(lib().dartk_optimize_poses_priors(C.byref(mo), mm, device._ptr(d_pri), arr, len(self.priors), device._ptr(vert),
                                                    device._ptr(norm), w, h, device._ptr(self.prediction), device._ptr(info),
                                                    device._ptr(first), device._stream()))
            back = (_abi.ContactPriorC * len(self.priors)).from_buffer_copy(d_pri.cpu().numpy().tobytes())
            for p, q in zip(self.priors, back):     # the reference moves the prior objects' contact points
                if p.update_contact_point:
                    p.contactPoint = (float(q.contactPoint[0]), float(q.contactPoint[1]), float(q.contactPoint[2]))
        else:
            check(lib().dartk_optimize_poses(C.byref(mo), mm, device._ptr(vert), device._ptr(norm), w, h, device._ptr(self.prediction),
                                             device._ptr(info), device._ptr(first), device._stream()))
        self._red_art_dev = red_art
        if not readback:
            return
        self.readPoses()
        if debug:
            self.iteration_summaries = info.cpu().numpy()
            self.first_system = first.cpu().numpy()
        return -1.0

    def readPoses(self):
        """Mirrors the poses the last optimizePosesDevice call left on the device into the host-side Pose objects (waits for
        the device)."""
        red_art = getattr(self, "_red_art_dev", None)
        if red_art is None:
            return
        for m in range(len(self.models)):      # the poses live on the device during the loop; mirror them back once
            T_mc = self.models[m].get_T_mc()
            p = self.poses[m]
            p.setTransformCameraToModel(T_mc)
            ra_n = p.getReducedArticulatedDimensions()
            if ra_n:
                p.getReducedArticulation()[:ra_n] = red_art[m].cpu().numpy()[:ra_n]
            p.projectReducedToFull()

    # ---- a batch of independent frames (BASELINE configs[4]), dartk_optimize_poses_batch
    def optimizePosesBatch(self, vert, norm, T_mcs, red_arts, debug=False):
        """Optimizer::optimizePoses for F independent frames in one library call. vert / norm: CUDA tensors [F,H,W,4];
        T_mcs: [F, nModels, 12] start transforms (numpy or CUDA tensor); red_arts: [F, sum of reduced articulation dims]
        (model order). Returns (T_mcs, red_arts) as numpy arrays (or tensors when tensors were passed); the tracker's own
        poses are not touched. Models, reductions, clouds and options are the tracker's."""
        import ctypes as C
        from . import _abi
        from ._abi import check, lib
        if self.priors:
            raise NotImplementedError("contact priors carry per-tracker state (their contact points): use optimizePosesDevice per frame")
        o, n = self.opts, len(self.models)
        dev = self.prediction.device
        F_, h, w = vert.shape[0], vert.shape[1], vert.shape[2]
        as_tensor = torch.is_tensor(T_mcs)
        T_d = (T_mcs if as_tensor else torch.from_numpy(np.ascontiguousarray(T_mcs, dtype=F))).to(dev).reshape(F_, n, 12).contiguous().clone()
        ra_total = int(sum(p.getReducedArticulatedDimensions() for p in self.poses))
        ra_d = (red_arts if torch.is_tensor(red_arts) else torch.from_numpy(np.ascontiguousarray(red_arts, dtype=F))).to(dev).reshape(F_, max(ra_total, 1))[:, :max(ra_total, 1)].contiguous().clone() \
            if ra_total else torch.zeros((F_, 1), dtype=torch.float32, device=dev)
        mm = (_abi.MultiModelC * n)()
        keep = []
        for m in range(n):
            p, red = self.poses[m], self.red_dev[m]
            ra_n = p.getReducedArticulatedDimensions()
            mins = np.asarray([p.getReducedMin(i) for i in range(ra_n)] or [0.0], dtype=F)
            maxs = np.asarray([p.getReducedMax(i) for i in range(ra_n)] or [0.0], dtype=F)
            b = np.ascontiguousarray(getattr(p._reduction, "_b", np.zeros(self.host_models[m].n_joints, dtype=F)), dtype=F)
            keep += [mins, maxs, b]
            c = mm[m]
            c.model = self.models[m].handle
            c.redType = 0 if red is None else (2 if red[0] == "map" else 1)
            c.redArtDims = ra_n
            c.d_A = device._ptr(red[1]) if red is not None and red[0] == "linear" else None
            c.d_map = device._ptr(red[1]) if red is not None and red[0] == "map" else None
            c.h_b, c.h_redMin, c.h_redMax = _abi.fptr(b), _abi.fptr(mins), _abi.fptr(maxs)
            c.d_redArt = None
            c.d_cloud, c.nCloud = device._ptr(self.clouds[m]), int(self.clouds[m].shape[0])
            c.d_potential = device._ptr(self.potentials[m])
            c.d_pts = None
            c.distThreshold, c.planeOffset = float(o.distThreshold[m]), float(o.planeOffset[m])
            for k in range(3):
                c.planeNormal[k] = float(o.planeNormal[m][k])
            c.regularization, c.regularizationScaled = float(o.regularization[m]), float(o.regularizationScaled[m])
            c.d_damping = self._damping_dev(m, keep)
        mo = _abi.MultiOptionsC()
        mo.numIterations, mo.nModels = int(o.numIterations), n
        mo.lambdaObsToMod, mo.lambdaModToObs = float(o.lambdaObsToMod), float(o.lambdaModToObs)
        mo.huberDelta, mo.focalLength = float(o.huberDelta), float(o.focalLength)
        mo.predWidth, mo.predHeight = self.pred_width, self.pred_height
        mo.predFl[0], mo.predFl[1] = float(self.pred_fl[0]), float(self.pred_fl[1])
        for i in range(n * n):
            mo.lambdaIntersection[i] = float(o.lambdaIntersection[i])
        info = torch.zeros((F_, max(o.numIterations, 1), n, 8), dtype=torch.float32, device=dev) if debug else None
        check(lib().dartk_optimize_poses_batch(C.byref(mo), mm, F_, device._ptr(vert), device._ptr(norm), w, h, device._ptr(T_d),
                                               device._ptr(ra_d) if ra_total else None, device._ptr(info), device._stream()))
        if debug:
            self.iteration_summaries = info.cpu().numpy()
        if as_tensor:
            return T_d, ra_d[:, :ra_total]
        return T_d.cpu().numpy(), ra_d[:, :ra_total].cpu().numpy()

    # ---- Optimizer::optimizePoses (optimizer.cpp:524-781), host-driven: one C-ABI call per term as the reference drives it
    def optimizePoses(self, vert, norm, debug=False):
        if self.priors:         # the priors are evaluated inside the device-resident loop's solve
            return self.optimizePosesDevice(vert, norm, debug=debug)
        o = self.opts
        n = len(self.models)
        models, poses = self.models, self.poses
        rdims = [p.getReducedDimensions() for p in poses]
        offs = np.concatenate([[0], np.cumsum(rdims)]).astype(int)
        D = int(offs[-1])
        info = np.zeros((max(o.numIterations, 1), n, 8), dtype=F)
        T_obs = []
        if o.lambdaModToObs > 0:                                  # generateObsSdf per model, :574-580
            for m in range(n):
                self.updatePose(m)
                device.generate_obs_sdf(vert, models[m], o.focalLength)
                T_obs.append(poses[m].getTransformCameraToModel().copy())
        for it in range(o.numIterations):
            U = np.zeros((D, D), dtype=F)
            JTe = np.zeros(D, dtype=F)
            for m in range(n):
                self.updatePose(m)
            if o.lambdaModToObs > 0:                              # raytracePrediction + cullUnobservable, :610-618
                device.raycast_prediction(models, self.pred_fl, (F(self.pred_width // 2), F(self.pred_height // 2)), self.pred_width,
                                          self.pred_height, prediction=self.prediction)
                device.cull_unobservable(self.prediction, vert)
            pts, counts, _ = device.error_and_data_association_multi(vert, norm, models, o.distThreshold[:n], o.planeOffset[:n],
                                                                     o.planeNormal[:n], pts_list=getattr(self, "_pts", None))
            self._pts = pts
            for m in range(n):
                dims, off, red = rdims[m], offs[m], self.red_dev[m]
                eJ = np.zeros(dims, dtype=F)
                J = np.zeros((dims, dims), dtype=F)
                eO2M = eM2O = eInt = F(0)
                n_pred = 0
                if o.lambdaObsToMod > 0:
                    s_ = device.norm_eqns_obs_to_mod(vert, models[m], pts[m], int(counts[m]), o.huberDelta, reduction=red).cpu().numpy()
                    eO2M = unpack(eJ, J, s_, o.lambdaObsToMod, dims)
                if o.lambdaModToObs > 0:
                    # reduced wrappers use the model's current T_mc, the full one T_obsSdf_c (modToObs.cu:851-988 vs :713-780)
                    s_d, cnt = device.norm_eqns_mod_to_obs(self.prediction, models[m], T_gc=T_obs[m], reduction=red)
                    eM2O = unpack(eJ, J, s_d.cpu().numpy(), o.lambdaModToObs, dims)
                    n_pred = int(cnt.item())
                lam_self = o.lambdaIntersection[m + m * n]
                if lam_self > 0 and dims > 6:
                    s_ = device.norm_eqns_self_intersection(self.clouds[m], models[m], self.potentials[m], reduction=red).cpu().numpy()
                    ad = dims - 6
                    eJt, Jt = np.zeros(ad, dtype=F), np.zeros((ad, ad), dtype=F)
                    eInt = unpack(eJt, Jt, s_, lam_self, ad)
                    eJ[6:] += eJt
                    J[6:, 6:] += Jt
                for d in range(n):
                    lam = o.lambdaIntersection[m + d * n]
                    if d == m or not lam > 0:
                        continue
                    T_ds = se3.mul(poses[d].getTransformCameraToModel(), poses[m].getTransformModelToCamera())
                    T_sd = se3.invert(T_ds)
                    s_ = device.norm_eqns_intersection(self.clouds[m], models[m], models[d], T_ds, T_sd, reduction=red).cpu().numpy()
                    eInt = unpack(eJ, J, s_, lam, dims)
                J = np.triu(J) + np.triu(J, 1).T                                   # make JTJ symmetric, :688-692
                J[np.diag_indices(dims)] += F(o.regularization[m]) * F(1.0)        # :695
                J[np.diag_indices(dims)] += F(o.regularizationScaled[m]) * J[np.diag_indices(dims)]     # :698-700
                J = (J + self.damping[m]).astype(F)                                # :702-703
                U[off:off + dims, off:off + dims] = np.triu(J)                     # :707-713
                JTe[off:off + dims] = eJ
                info[it, m] = (eO2M, eM2O, eInt, counts[m], n_pred, 0, 0, 0)
            if it == 0 and debug:
                self.first_system = np.concatenate([U.reshape(-1), JTe])
            x = (-solve_upper(U, JTe)).astype(F)                                   # :732
            for m in range(n):                                                     # :734-755
                p, off = poses[m], offs[m]
                dT = se3.exp(x[off:off + 6])
                p.setTransformCameraToModel(se3.mul(dT, p.getTransformCameraToModel()))
                ra = p.getReducedArticulation()
                for i in range(p.getReducedArticulatedDimensions()):
                    lo, hi = (p.getReducedMin(i), p.getReducedMax(i))
                    ra[i] = min(max(F(lo), F(ra[i] + x[off + 6 + i])), F(hi))
                self.updatePose(m)
        self.iteration_summaries = info
        return -1.0
