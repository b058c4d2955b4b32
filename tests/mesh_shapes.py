"""Synthetic triangle meshes for the mesh-pipeline tests (the reference's mesh assets are not in its tree, and its one
in-tree mesh cannot travel to the GPU box): closed, consistently wound surfaces built here."""
import numpy as np

F = np.float32


def icosphere(radius=0.05, subdivisions=2, center=(0, 0, 0)):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.asarray(v) * radius + np.asarray(center)).astype(F), np.asarray(f, dtype=np.int32)


def box(size=(0.06, 0.04, 0.05), center=(0, 0, 0)):
    """12 axis-aligned triangles: every face hits the zero-thickness branch of the candidate box."""
    s = np.asarray(size, dtype=np.float64) / 2
    v = np.asarray([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], dtype=np.float64) * s + np.asarray(center)
    f = [(0, 2, 3), (0, 3, 1), (4, 5, 7), (4, 7, 6), (0, 1, 5), (0, 5, 4), (2, 6, 7), (2, 7, 3), (0, 4, 6), (0, 6, 2), (1, 3, 7), (1, 7, 5)]
    return v.astype(F), np.asarray(f, dtype=np.int32)


def torus(R=0.05, r=0.018, nu=24, nv=12, center=(0, 0, 0)):
    v, f = [], []
    for a in range(nu):
        for b in range(nv):
            u, w = 2 * np.pi * a / nu, 2 * np.pi * b / nv
            v.append(((R + r * np.cos(w)) * np.cos(u), (R + r * np.cos(w)) * np.sin(u), r * np.sin(w)))
    for a in range(nu):
        for b in range(nv):
            p00, p10 = a * nv + b, ((a + 1) % nu) * nv + b
            p01, p11 = a * nv + (b + 1) % nv, ((a + 1) % nu) * nv + (b + 1) % nv
            f += [(p00, p10, p11), (p00, p11, p01)]
    return (np.asarray(v) + np.asarray(center)).astype(F), np.asarray(f, dtype=np.int32)


def grid_around(vertices, resolution=0.005, padding=0.02):
    """dim / offset as voxelizeFrame derives them (host_only_model.cpp:639-661)."""
    mn = (vertices.min(axis=0) - F(padding)).astype(F)
    mx = (vertices.max(axis=0) + F(padding)).astype(F)
    dim = np.ceil((mx - mn) / F(resolution)).astype(np.int64)
    return tuple(int(d) for d in dim), mn
