"""Mesh side of the model/asset pipeline: triangle-mesh import, the scale/transform applied to mesh
geometry, solid-mesh voxel splat (on the device), area-weighted surface sampling for collision clouds, and the SDF
cache file format. Cold path: runs once per model, never per Gauss-Newton iteration.

Reference behaviour mirrored here:
  * mesh/assimp_mesh_reader.cpp:12-52 -- one triangle mesh per file, non-triangle faces are an error. The reference
    delegates parsing to assimp (not in this image, not vendored by the reference); the readers below parse OBJ,
    ASCII/binary STL and ASCII PLY directly. Only positions and faces are used downstream (normals feed the GUI
    renderer, out of scope), so assimp's per-corner vertex duplication makes no difference to any result;
  * mesh/mesh_proc.cpp (scaleMesh / transformMesh) -- per-axis scale, then the geometry's SE3;
  * mesh/mesh_splat.cpp:10-265 -- `splat_solid_mesh`, one call into libdartk (dartk_splat_solid_mesh);
  * mesh/mesh_sample.cpp:13-72 -- `sample_mesh`, with glibc's rand() restated (`GlibcRand`; the reference never seeds,
    i.e. seed 1), pinned by tests/golden/glibc_rand.json;
  * model/host_only_model.cpp:273-325 -- `sdf_cache_filename`, `write_sdf_cache`, `read_sdf_cache`.
"""
import ctypes as C
import os
import struct

import numpy as np

F = np.float32


class Mesh:
    """dart::Mesh (mesh/mesh.h): float3 vertices, int3 faces."""

    def __init__(self, vertices, faces):
        self.vertices = np.ascontiguousarray(vertices, dtype=F).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        if len(self.faces) and (self.faces.min() < 0 or self.faces.max() >= len(self.vertices)):
            raise ValueError("mesh face indexes a vertex outside [0,%d)" % len(self.vertices))

    @property
    def n_vertices(self):
        return len(self.vertices)

    @property
    def n_faces(self):
        return len(self.faces)

    def scaled(self, scale):
        """scaleMesh (mesh_proc.cpp): component-wise."""
        s = np.asarray(scale, dtype=F).reshape(1, 3)
        return Mesh((self.vertices * s).astype(F), self.faces)

    def transformed(self, T):
        """transformMesh (mesh_proc.cpp): SE3Transform of every vertex, fp32, in the order of SE3.h:435-439."""
        T = np.asarray(T, dtype=F).reshape(3, 4)
        v = self.vertices
        out = np.empty_like(v)
        for r in range(3):
            out[:, r] = (((T[r, 0] * v[:, 0]).astype(F) + (T[r, 1] * v[:, 1]).astype(F)).astype(F) +
                         (T[r, 2] * v[:, 2]).astype(F)).astype(F) + T[r, 3]
        return Mesh(out.astype(F), self.faces)

    def bounds(self):
        return self.vertices.min(axis=0), self.vertices.max(axis=0)


# ---- readers ------------------------------------------------------------------------------------------------------
def _face_index(token, n):
    i = int(token.split("/")[0])
    return i - 1 if i > 0 else n + i


def read_obj(path):
    verts, faces = [], []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
            elif parts[0] == "f":
                if len(parts) != 4:
                    raise ValueError("%s is not a triangle mesh" % path)     # assimp_mesh_reader.cpp:33-38
                faces.append([_face_index(t, len(verts)) for t in parts[1:4]])
    if not faces:
        raise ValueError("%s holds no faces" % path)
    return Mesh(verts, faces)


def read_stl(path):
    with open(path, "rb") as fh:
        raw = fh.read()
    n_bin = struct.unpack_from("<I", raw, 80)[0] if len(raw) >= 84 else -1
    if n_bin >= 0 and len(raw) == 84 + 50 * n_bin:                          # binary: 80-byte header, count, 50-byte records
        rec = np.frombuffer(raw, dtype=np.uint8, count=50 * n_bin, offset=84).reshape(n_bin, 50)
        tri = rec[:, 12:48].copy().view("<f4").reshape(n_bin, 3, 3)
    else:
        vals = [ln.split()[1:4] for ln in raw.decode("ascii", "replace").splitlines() if ln.strip().startswith("vertex")]
        if len(vals) % 3:
            raise ValueError("%s is not a triangle mesh" % path)
        tri = np.asarray(vals, dtype=np.float64).reshape(-1, 3, 3)
    if len(tri) == 0:
        raise ValueError("%s holds no faces" % path)
    return Mesh(tri.reshape(-1, 3), np.arange(3 * len(tri)).reshape(-1, 3))


def read_ply(path):
    with open(path, "r", errors="replace") as fh:
        lines = fh.read().splitlines()
    if not lines or lines[0].strip() != "ply":
        raise ValueError("%s is not a PLY file" % path)
    nv = nf = 0
    body = None
    for k, ln in enumerate(lines):
        t = ln.split()
        if t[:2] == ["format", "ascii"]:
            continue
        if t[:1] == ["format"]:
            raise ValueError("%s: only ASCII PLY is read here" % path)
        if t[:2] == ["element", "vertex"]:
            nv = int(t[2])
        elif t[:2] == ["element", "face"]:
            nf = int(t[2])
        elif t[:1] == ["end_header"]:
            body = k + 1
            break
    if body is None:
        raise ValueError("%s: no end_header" % path)
    verts = [[float(x) for x in lines[body + i].split()[:3]] for i in range(nv)]
    faces = []
    for i in range(nf):
        t = lines[body + nv + i].split()
        if int(t[0]) != 3:
            raise ValueError("%s is not a triangle mesh" % path)
        faces.append([int(t[1]), int(t[2]), int(t[3])])
    return Mesh(verts, faces)


def read_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    reader = {".obj": read_obj, ".stl": read_stl, ".ply": read_ply}.get(ext)
    if reader is None:
        raise ValueError("no reader for mesh file %r (obj, stl, ply)" % path)
    return reader(path)


def write_obj(path, mesh):
    with open(path, "w") as fh:
        for v in mesh.vertices:
            fh.write("v %.9g %.9g %.9g\n" % tuple(float(c) for c in v))
        for f in mesh.faces:
            fh.write("f %d %d %d\n" % tuple(int(i) + 1 for i in f))


# ---- solid-mesh voxel splat (device) ----------------------------------------------------------------------------------
def splat_solid_mesh(mesh, dim, offset, resolution, data, background=None):
    """splatSolidMesh on the GPU: returns `data` (float32 [z][y][x], or None to start from `background`) with 0 written
    into every voxel inside the mesh. Raises when libdartk is missing: there is no CPU path."""
    import torch
    from . import _abi
    lib = _abi.lib()
    dim = tuple(int(v) for v in dim)
    if data is None:
        data = np.full((dim[2], dim[1], dim[0]), F(1e20 if background is None else background), dtype=F)
    data = np.ascontiguousarray(data, dtype=F).reshape(dim[2], dim[1], dim[0])
    if mesh.n_faces == 0:
        return data
    dev = torch.device("cuda", torch.cuda.current_device())
    d_v = torch.from_numpy(mesh.vertices.copy()).to(dev)
    d_f = torch.from_numpy(mesh.faces.copy()).to(dev)
    d_s = torch.from_numpy(data.copy()).to(dev)
    off = (C.c_float * 3)(*[float(v) for v in np.asarray(offset, dtype=F)])
    st = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.dartk_splat_solid_mesh(d_v.data_ptr(), mesh.n_vertices, d_f.data_ptr(), mesh.n_faces, dim[0], dim[1], dim[2],
                                          off, C.c_float(float(F(resolution))), d_s.data_ptr(), st))
    return d_s.cpu().numpy()


# ---- glibc rand() -------------------------------------------------------------------------------------------------------
class GlibcRand:
    """glibc's default rand(): the TYPE_3 additive feedback generator of random_r.c (degree 31, separation 3), seeded
    by srand(seed) with the 16807 Lehmer sequence and 310 discarded outputs. RAND_MAX = 2**31 - 1."""
    RAND_MAX = 2147483647

    def __init__(self, seed=1):
        seed = int(seed) & 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            word = r[i - 1] if r[i - 1] < 2 ** 31 else r[i - 1] - 2 ** 32      # int32 arithmetic
            hi, lo = int(word / 127773), int(np.fmod(word, 127773))              # C division truncates toward zero
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word & 0xFFFFFFFF
        self.state = r[:31]
        self.f, self.r = 3, 0
        for _ in range(310):
            self._step()

    def _step(self):
        s = self.state
        s[self.f] = (s[self.f] + s[self.r]) & 0xFFFFFFFF
        out = s[self.f] >> 1
        self.f = (self.f + 1) % 31
        self.r = (self.r + 1) % 31
        return out

    def rand(self):
        return self._step()

    def draws(self, n):
        return np.fromiter((self._step() for _ in range(n)), dtype=np.int64, count=n)


def sample_mesh(mesh, sample_density, rng=None):
    """sampleMesh (mesh_sample.cpp:13-72): round(area * density) points, each from a face drawn with probability
    proportional to its (Heron) area and uniform barycentric coordinates; three rand() calls per point."""
    rng = rng or GlibcRand(1)
    if mesh.n_faces == 0:
        return np.zeros((0, 3), dtype=F)
    V, Fc = mesh.vertices, mesh.faces
    A, B, Cc = V[Fc[:, 0]], V[Fc[:, 1]], V[Fc[:, 2]]

    def length(d):      # helper_math length(): sqrtf(x*x + y*y + z*z) in fp32
        return np.sqrt(((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F) + (d[:, 2] * d[:, 2]).astype(F)).astype(F)

    a = length(A - B).astype(np.float64)
    b = length(B - Cc).astype(np.float64)
    c = length(Cc - A).astype(np.float64)
    s = (a + b + c) / 2
    with np.errstate(invalid="ignore"):
        area = np.sqrt(s * (s - a) * (s - b) * (s - c))
    area[np.isnan(area)] = 0
    cum = np.empty(len(area), dtype=np.float64)     # serial running sum, as written there
    acc = 0.0
    for i, v in enumerate(area):
        acc = v if i == 0 else acc + v
        cum[i] = acc
    total = cum[-1]
    n = int(np.floor(total * float(F(sample_density)) + 0.5)) if total * float(F(sample_density)) >= 0 else 0   # round()
    if n == 0:
        return np.zeros((0, 3), dtype=F)
    d = rng.draws(3 * n).reshape(n, 3)
    r0 = d[:, 0].astype(np.float64) * total / GlibcRand.RAND_MAX
    face = np.searchsorted(cum, r0, side="left")                                 # std::lower_bound
    r1 = (d[:, 1].astype(F) / F(GlibcRand.RAND_MAX)).astype(F)                   # int -> float, then fp32 division
    r2 = (d[:, 2].astype(F) / F(GlibcRand.RAND_MAX)).astype(F)
    q = np.sqrt(r1).astype(F)
    wA = (F(1) - q).astype(F)[:, None]
    wB = (q * (F(1) - r2).astype(F)).astype(F)[:, None]
    wC = (q * r2).astype(F)[:, None]
    Af, Bf, Cf = A[face], B[face], Cc[face]
    return (((wA * Af).astype(F) + (wB * Bf).astype(F)).astype(F) + (wC * Cf).astype(F)).astype(F)


# ---- SDF cache files (host_only_model.cpp:273-325) ------------------------------------------------------------------------
def sdf_cache_filename(cache_file, sdf_number, resolution):
    return "%s.sdf%02d.res%06f.sdf" % (cache_file, sdf_number, float(F(resolution)))


def write_sdf_cache(path, dim, offset, resolution, data):
    data = np.ascontiguousarray(data, dtype="<f4")
    with open(path, "wb") as fh:
        fh.write(struct.pack("<3i", *[int(v) for v in dim]))
        fh.write(struct.pack("<3f", *[float(v) for v in offset]))
        fh.write(struct.pack("<f", float(resolution)))
        fh.write(data.tobytes())


def read_sdf_cache(path):
    with open(path, "rb") as fh:
        raw = fh.read()
    dim = struct.unpack_from("<3i", raw, 0)
    offset = np.asarray(struct.unpack_from("<3f", raw, 12), dtype=F)
    resolution = F(struct.unpack_from("<f", raw, 24)[0])
    n = dim[0] * dim[1] * dim[2]
    if min(dim) <= 0 or len(raw) != 28 + 4 * n:
        raise ValueError("%s: not an SDF cache file (dim %r, %d bytes)" % (path, dim, len(raw)))
    data = np.frombuffer(raw, dtype="<f4", count=n, offset=28).reshape(dim[2], dim[1], dim[0]).copy()
    return dim, offset, resolution, data
