"""Pose and pose reductions (src/pose/pose.h, pose_reduction.h, pose.cpp) -- host-side mirror.

A pose is T_mc/T_cm plus a full articulation vector and a reduced one; NullReduction aliases the two
(pose.cpp:9-10), LinearPoseReduction maps full = b + A*reduced with A[r + f*redDims] (pose.cpp:66-73) and
ParamMapPoseReduction maps full[f] = reduced[mapping[f]] (pose.cpp:91-97). Dimension getters include the 6 rigid
DOF exactly as the reference's do (pose.h:27-33).
"""
import numpy as np

from . import se3

F = np.float32


class PoseReduction:
    def __init__(self, full_dims, red_dims, mins=None, maxs=None, names=None):
        self._full, self._red = int(full_dims), int(red_dims)
        self._mins = np.zeros(red_dims, dtype=F) if mins is None else np.asarray(mins, dtype=F).copy()
        self._maxs = np.zeros(red_dims, dtype=F) if maxs is None else np.asarray(maxs, dtype=F).copy()
        self._names = [""] * red_dims if names is None else list(names)

    def getFullDimensions(self):
        return self._full

    def getReducedDimensions(self):
        return self._red

    def getMin(self, d):
        return self._mins[d]

    def getMax(self, d):
        return self._maxs[d]

    def getName(self, d):
        return self._names[d]

    def isNull(self):
        return False

    def isParamMap(self):
        return False


class NullReduction(PoseReduction):
    def __init__(self, dims, mins=None, maxs=None, names=None):
        super().__init__(dims, dims, mins, maxs, names)

    def isNull(self):
        return True

    def projectReducedToFull(self, reduced, full):
        pass

    def getFirstDerivatives(self):
        return None


class LinearPoseReduction(PoseReduction):
    def __init__(self, full_dims, red_dims, A=None, b=None, mins=None, maxs=None, names=None):
        super().__init__(full_dims, red_dims, mins, maxs, names)
        self._A = np.zeros(full_dims * red_dims, dtype=F) if A is None else np.asarray(A, dtype=F).reshape(-1).copy()
        self._b = np.zeros(full_dims, dtype=F) if b is None else np.asarray(b, dtype=F).copy()
        assert self._A.size == full_dims * red_dims

    def projectReducedToFull(self, reduced, full):
        A = self._A.reshape(self._full, self._red)
        for f in range(self._full):
            acc = F(self._b[f])
            for r in range(self._red):
                acc = F(acc + F(reduced[r] * A[f, r]))
            full[f] = acc

    def getFirstDerivatives(self):
        return self._A

    def getFirstDerivative(self, reduced_dim, full_dim):
        return self._A[reduced_dim + full_dim * self._red]


class ParamMapPoseReduction(LinearPoseReduction):
    def __init__(self, full_dims, red_dims, mapping, mins=None, maxs=None, names=None):
        mapping = np.asarray(mapping, dtype=np.int32).copy()
        A = np.zeros((full_dims, red_dims), dtype=F)
        A[np.arange(full_dims), mapping] = 1
        super().__init__(full_dims, red_dims, A, np.zeros(full_dims, dtype=F), mins, maxs, names)
        self._mapping = mapping

    def projectReducedToFull(self, reduced, full):
        for f in range(self._full):
            full[f] = reduced[self._mapping[f]]

    def getMapping(self):
        return self._mapping

    def isParamMap(self):
        return True


class Pose:
    def __init__(self, reduction):
        self._reduction = reduction
        self._full = np.zeros(reduction.getFullDimensions(), dtype=F)
        self._reduced = self._full if reduction.isNull() else np.zeros(reduction.getReducedDimensions(), dtype=F)
        self._T_mc = se3.identity()
        self._T_cm = se3.identity()

    def copy(self):
        p = Pose(self._reduction)
        p._reduced[:] = self._reduced
        p._full[:] = self._full
        p._T_mc, p._T_cm = self._T_mc.copy(), self._T_cm.copy()
        return p

    def getDimensions(self):
        return self._reduction.getFullDimensions() + 6

    def getReducedDimensions(self):
        return self._reduction.getReducedDimensions() + 6

    def getArticulatedDimensions(self):
        return self._reduction.getFullDimensions()

    def getReducedArticulatedDimensions(self):
        return self._reduction.getReducedDimensions()

    def getArticulation(self):
        return self._full

    def getReducedArticulation(self):
        return self._reduced

    def getReducedMin(self, d):
        return self._reduction.getMin(d)

    def getReducedMax(self, d):
        return self._reduction.getMax(d)

    def getReducedName(self, d):
        return self._reduction.getName(d)

    def isReduced(self):
        return not self._reduction.isNull()

    def getReduction(self):
        return self._reduction

    def getFirstDerivatives(self):
        return self._reduction.getFirstDerivatives()

    def zero(self):
        self._reduced[:] = 0
        self._T_mc, self._T_cm = se3.identity(), se3.identity()

    def projectReducedToFull(self):
        self._reduction.projectReducedToFull(self._reduced, self._full)

    def getTransformCameraToModel(self):
        return self._T_mc

    def getTransformModelToCamera(self):
        return self._T_cm

    def setTransformCameraToModel(self, T_mc):
        self._T_mc = np.asarray(T_mc, dtype=F).reshape(12).copy()
        self._T_cm = se3.invert(self._T_mc)

    def setTransformModelToCamera(self, T_cm):
        self._T_cm = np.asarray(T_cm, dtype=F).reshape(12).copy()
        self._T_mc = se3.invert(self._T_cm)


def null_pose_for(host_model):
    lim = np.asarray(host_model.joint_limits, dtype=F).reshape(-1, 2)
    return Pose(NullReduction(host_model.n_joints, lim[:, 0], lim[:, 1], host_model.joint_names))
