"""Spatial ordering of point sets handed to the device path.

The intersection and contact-prior kernels walk a point's destination SDFs one wavefront (64 consecutive points) at a
time and drop a (point, SDF) pair early when the point is far from the SDF; the early exit only pays when the 64 points of
a wavefront agree, i.e. when consecutive points are close in space. The reference samples its collision clouds with
rand() (tracker.cpp:143-210, mesh_sample.cpp:13-72), so the order of the points carries no meaning there; ordering them
along a Z-order curve changes nothing but the order in which the (commutative) normal-equation sums are formed.
"""
import numpy as np


def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton_order(xyz, bits=10):
    """Stable permutation that sorts the points [n, 3] along the Z-order curve of their bounding box."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    if len(xyz) == 0:
        return np.zeros(0, dtype=np.int64)
    lo = xyz.min(0)
    span = np.maximum(xyz.max(0) - lo, 1e-12).max()
    q = np.minimum(((xyz - lo) / span * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << np.uint64(1)) | (_spread3(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")
