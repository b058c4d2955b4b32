"""Seeded synthetic inputs shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np

from dart_amd import se3
from dart_amd.model import load_asset, sphere_model

F = np.float32
_cache = {}


def sphere(resolution=0.005, padding=0.1, radius=0.1):
    key = ("sphere", resolution, padding, radius)
    if key not in _cache:
        _cache[key] = sphere_model(radius).voxelize2(resolution, padding)
    return _cache[key]


def articulated(name, resolution=0.005, padding=0.10):
    key = (name, resolution, padding)
    if key not in _cache:
        _cache[key] = load_asset(name).voxelize(resolution, padding)
    return _cache[key]


def kat1_vertex_map(order="as_compiled"):
    """The reference kernel test's OWN observation map (test_obs_to_mod_kernels.cpp:42-53): 10x10, drawn from libstdc++'s
    std::default_random_engine / bernoulli_distribution(0.8) / normal_distribution<float>(0, 0.1) by the committed generator
    tests/golden/gen_kat1_inputs.cpp (fixture tests/golden/kat1_inputs.json, floats stored as bit patterns). `order`: which
    argument evaluation order filled x, y, z ("as_compiled" = g++, the reference's host compiler; "left_to_right")."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_inputs.json")) as fh:
        d = json.load(fh)
    v = np.array(d[order], dtype=np.uint32).view(F)
    return np.ascontiguousarray(v.reshape(d["height"], d["width"], 4))


def random_vertex_map(w, h, seed, sigma=0.1, p_valid=0.8, center=(0, 0, 0)):
    """Larger maps with the statistics of the reference kernel test's observation (points ~ N(0, sigma^2) per axis, 80 %
    valid) from a seeded numpy generator; the reference's own 10x10 draw is kat1_vertex_map()."""
    rng = np.random.default_rng(seed)
    v = np.zeros((h, w, 4), dtype=F)
    valid = rng.random((h, w)) < p_valid
    v[..., :3] = rng.normal(0, sigma, size=(h, w, 3)).astype(F) + np.asarray(center, dtype=F)
    v[..., 3] = 1
    v[~valid] = 0
    return v


def random_normal_map(w, h, seed, p_valid=0.7):
    rng = np.random.default_rng(seed + 1000)
    n = rng.normal(size=(h, w, 3))
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    out = np.zeros((h, w, 4), dtype=F)
    out[..., :3] = n.astype(F)
    out[..., 3] = (rng.random((h, w)) < p_valid).astype(F)
    out[out[..., 3] == 0] = 0
    return out


def random_joints(model, seed, margin=1e-3):
    rng = np.random.default_rng(seed)
    lim = np.asarray(model.joint_limits, dtype=np.float64).reshape(-1, 2)
    return (lim[:, 0] + margin + (lim[:, 1] - lim[:, 0] - 2 * margin) * rng.random(len(lim))).astype(F)


def random_T(seed, trans=0.05, rot=0.3, base=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    t6 = np.concatenate([rng.uniform(-trans, trans, 3) + np.asarray(base), rng.uniform(-rot, rot, 3)]).astype(F)
    return se3.exp(t6)


def points_near_model(model, T_mf, T_cm, w, h, seed, sigma=0.01, p_valid=0.8):
    """Vertex map whose valid points lie near the surfaces of the posed model: uniform samples inside each SDF's
    un-padded box mapped to the camera frame. Exercises every SDF / frame."""
    rng = np.random.default_rng(seed)
    n = w * h
    s = rng.integers(0, len(model.sdfs), n)
    pts = np.zeros((n, 3), dtype=np.float64)
    for k in range(len(model.sdfs)):
        g = model.sdfs[k]
        sel = np.where(s == k)[0]
        lo = g.offset.astype(np.float64) + 0.07
        hi = g.offset.astype(np.float64) + g.resolution * np.asarray(g.dim) - 0.07
        p = lo + (hi - lo) * rng.random((len(sel), 3))
        pm = se3.apply(T_mf[model.sdf_frame[k]], p.astype(F))
        pts[sel] = se3.apply(T_cm, pm.astype(F))
    v = np.zeros((h, w, 4), dtype=F)
    v[..., :3] = pts.reshape(h, w, 3).astype(F)
    v[..., 3] = 1
    v[rng.random((h, w)) >= p_valid] = 0
    return v
