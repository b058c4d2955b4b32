"""Mesh side of the asset pipeline on the CPU: glibc rand() restatement against golden vectors and against libc itself,
sampleMesh against the oracle bit for bit, mesh readers, SDF cache files. Parity unpinned for sampleMesh (the
reference has no test for it): the oracle restates mesh_sample.cpp:13-72 and draws from the real libc rand()."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
import mesh_shapes as S
from dart_amd import mesh as M

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def test_glibc_rand_matches_golden_and_libc():
    with open(os.path.join(HERE, "golden", "glibc_rand.json")) as fh:
        g = json.load(fh)
    assert g["rand_max"] == M.GlibcRand.RAND_MAX
    for seed, draws in g["draws"].items():
        assert np.array_equal(M.GlibcRand(int(seed)).draws(len(draws)), np.asarray(draws))
    for seed in (1, 7, 99991):
        assert np.array_equal(M.GlibcRand(seed).draws(5000), O.libc_rand(seed, 5000))


@pytest.mark.parametrize("shape", ["icosphere", "box", "torus"])
def test_sample_mesh_bit_exact_with_oracle(shape):
    v, f = getattr(S, shape)()
    for density, seed in ((2e4, 1), (1e5, 3)):
        got = M.sample_mesh(M.Mesh(v, f), density, M.GlibcRand(seed))
        want = O.sample_mesh(v, f, density, seed)
        assert got.shape == want.shape and len(got) > 50
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the samples lie on the surface: inside the bounding box, and for the sphere within the facet sagitta of the radius
    assert (got >= v.min(axis=0) - 1e-6).all() and (got <= v.max(axis=0) + 1e-6).all()
    if shape == "icosphere":
        r = np.linalg.norm(got, axis=1)
        assert r.max() <= 0.05 + 1e-6 and r.min() > 0.05 * 0.97


def test_sample_mesh_degenerate_faces_and_empty():
    v = np.asarray([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0]], dtype=F)       # first face has zero area
    f = np.asarray([[0, 1, 2], [0, 1, 3]], dtype=np.int32)
    got = M.sample_mesh(M.Mesh(v, f), 100.0, M.GlibcRand(1))
    want = O.sample_mesh(v, f, 100.0, 1)
    assert len(got) == 50 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert len(M.sample_mesh(M.Mesh(v, f), 0.0)) == 0
    assert len(M.sample_mesh(M.Mesh(v, np.zeros((0, 3), np.int32)), 10.0)) == 0


def test_mesh_readers_round_trip(tmp_path):
    v, f = S.icosphere(subdivisions=1)
    mesh = M.Mesh(v, f)
    p = str(tmp_path / "a.obj")
    M.write_obj(p, mesh)
    got = M.read_mesh(p)
    assert np.array_equal(got.vertices, v) and np.array_equal(got.faces, f)
    # OBJ with texture/normal indices and negative (relative) indices
    with open(str(tmp_path / "b.obj"), "w") as fh:
        fh.write("# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1\nf -3//1 -2//1 -1//1\n")
    b = M.read_mesh(str(tmp_path / "b.obj"))
    assert b.faces.tolist() == [[0, 1, 2], [0, 1, 2]]
    with open(str(tmp_path / "quad.obj"), "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(ValueError, match="not a triangle mesh"):           # assimp_mesh_reader.cpp:33-38
        M.read_mesh(str(tmp_path / "quad.obj"))
    # STL, ASCII and binary: unshared corners, same triangles
    tri = v[f]
    with open(str(tmp_path / "a.stl"), "w") as fh:
        fh.write("solid s\n")
        for t in tri:
            fh.write("facet normal 0 0 0\nouter loop\n" + "".join("vertex %.9g %.9g %.9g\n" % tuple(map(float, c)) for c in t) + "endloop\nendfacet\n")
        fh.write("endsolid s\n")
    with open(str(tmp_path / "b.stl"), "wb") as fh:
        fh.write(b"\0" * 80 + struct.pack("<I", len(tri)))
        for t in tri:
            fh.write(struct.pack("<12fH", 0, 0, 0, *[float(c) for c in t.reshape(-1)], 0))
    for name in ("a.stl", "b.stl"):
        s = M.read_mesh(str(tmp_path / name))
        assert s.n_faces == len(f) and np.array_equal(s.vertices[s.faces], tri)
    with open(str(tmp_path / "a.ply"), "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(f)))
        fh.write("".join("%.9g %.9g %.9g\n" % tuple(map(float, c)) for c in v))
        fh.write("".join("3 %d %d %d\n" % tuple(map(int, c)) for c in f))
    pl = M.read_mesh(str(tmp_path / "a.ply"))
    assert np.array_equal(pl.vertices, v) and np.array_equal(pl.faces, f)
    with pytest.raises(ValueError):
        M.read_mesh(str(tmp_path / "a.xyz"))
    with pytest.raises(ValueError, match="outside"):
        M.Mesh(v, [[0, 1, len(v)]])


def test_scale_and_transform_order():
    v, f = S.box()
    T = np.asarray([0, -1, 0, 0.1, 1, 0, 0, 0.2, 0, 0, 1, 0.3], dtype=F)
    m = M.Mesh(v, f).scaled([2, 1, 0.5]).transformed(T)
    want = np.stack([-(v[:, 1] * F(1)) + F(0.1), v[:, 0] * F(2) + F(0.2), v[:, 2] * F(0.5) + F(0.3)], axis=1).astype(F)
    assert np.allclose(m.vertices, want, atol=1e-7)
    lo, hi = m.bounds()
    assert (lo < hi).all()


def test_sdf_cache_format(tmp_path):
    assert M.sdf_cache_filename("/tmp/hand", 3, 0.005) == "/tmp/hand.sdf03.res0.005000.sdf"      # "%s.sdf%02d.res%06f.sdf"
    rng = np.random.default_rng(0)
    data = rng.normal(size=(5, 4, 3)).astype(F)      # [z][y][x]
    p = str(tmp_path / "m.sdf00.res0.005000.sdf")
    M.write_sdf_cache(p, (3, 4, 5), (0.1, -0.2, 0.3), 0.005, data)
    raw = open(p, "rb").read()
    assert len(raw) == 12 + 12 + 4 + 4 * 60         # int3 dim, float3 offset, float resolution, data (host_only_model.cpp:317-325)
    assert struct.unpack_from("<3i", raw, 0) == (3, 4, 5)
    dim, off, res, got = M.read_sdf_cache(p)
    assert dim == (3, 4, 5) and np.allclose(off, [0.1, -0.2, 0.3]) and res == F(0.005)
    assert np.array_equal(got, data)
    with open(p, "ab") as fh:
        fh.write(b"x")
    with pytest.raises(ValueError):
        M.read_sdf_cache(p)


def test_model_description_keeps_mesh_geometry(tmp_path):
    """A model read from XML with a mesh geom survives the JSON round trip (the mesh file is named, not embedded)."""
    import json
    from dart_amd.model import HostModel, read_model_xml, GEOM_MESH
    v, f = S.icosphere(radius=1.0, subdivisions=1)
    M.write_obj(str(tmp_path / "ball.obj"), M.Mesh(v, f))
    with open(str(tmp_path / "m.xml"), "w") as fh:
        fh.write('<?xml version="1.0" ?>\n<model version="1">\n  <geom type="mesh" sx="0.03" sy="0.03" sz="0.03" tx="0" ty="0" tz="0" '
                 'rx="0" ry="0" rz="0" red="0" green="0" blue="0" meshFile="ball.obj"/>\n</model>\n')
    m = read_model_xml(str(tmp_path / "m.xml"), load_meshes=True)
    assert m.geoms[0]["type"] == GEOM_MESH and m.geoms[0]["mesh"].n_faces == len(f)
    lo, hi = m._frame_extent(m.frame_geoms(0))
    assert np.allclose(lo, -0.03, atol=1e-6) and np.allclose(hi, 0.03, atol=1e-6)
    spec = m.to_json()
    assert spec["geoms"][0]["meshFile"] == "ball.obj"
    with open(str(tmp_path / "m.json"), "w") as fh:
        json.dump(spec, fh)
    m2 = HostModel.from_json(str(tmp_path / "m.json"))
    assert m2.geoms[0]["mesh"].n_faces == len(f) and np.array_equal(m2.geoms[0]["mesh"].vertices, m.geoms[0]["mesh"].vertices)
    spec["geoms"][0].pop("meshFile")
    with pytest.raises(ValueError):
        HostModel.from_json(spec)
    # without load_meshes and without a proxy the mesh geom is dropped, as before
    assert len(read_model_xml(str(tmp_path / "m.xml")).geoms) == 0
