"""ImageDepthSource (dart_amd/depth_source.py): 16-bit PNG sequences -> depth -> vertex/normal maps."""
import numpy as np
import pytest
from PIL import Image

from dart_amd.depth_source import ImageDepthSource


def write_sequence(d, n=3, w=64, h=48, seed=0):
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(n):
        a = rng.integers(300, 1500, size=(h, w)).astype(np.uint16)
        a[rng.random((h, w)) < 0.2] = 0
        Image.fromarray(a).save(str(d / ("depth_%04d.png" % (k + 5))))
        frames.append(a)
    return frames


def test_directory_and_format_initialisers(tmp_path):
    frames = write_sequence(tmp_path)
    (tmp_path / "notes.txt").write_text("x")
    s = ImageDepthSource()
    assert s.initialize_directory(str(tmp_path), (525.0, 525.0), scale_to_meters=0.001)
    assert s.getNumDepthFrames() == 3 and (s.getDepthWidth(), s.getDepthHeight()) == (64, 48)
    assert s.principal_point == (32.0, 24.0)
    assert np.array_equal(s.getDepth(), frames[0])
    s.advance(); assert np.array_equal(s.getDepth(), frames[1])
    s.advance(); s.advance()                                   # wraps to frame 0
    assert s.getFrame() == 0 and np.array_equal(s.getDepth(), frames[0])
    s.setFrame(2); assert np.array_equal(s.getDepth(), frames[2])
    s.setPngSwap(False); s.setFrame(2)
    assert np.array_equal(s.getDepth(), frames[2].byteswap())
    f = ImageDepthSource()
    assert f.initialize_format(str(tmp_path / "depth_%04d.png"), (525.0, 525.0), (31.5, 23.5), 64, 48, first_depth_frame=6)
    assert f.getNumDepthFrames() == 2 and np.array_equal(f.getDepth(), frames[1]) and f.principal_point == (31.5, 23.5)
    assert not ImageDepthSource().initialize_directory(str(tmp_path / "missing"), (1, 1))
    # the default principal point: integer division of an odd size, and only principalPoint.x is tested
    # (image_depth_source.h:227-231)
    odd = tmp_path / "odd"
    odd.mkdir()
    write_sequence(odd, n=1, w=63, h=47)
    o = ImageDepthSource()
    assert o.initialize_directory(str(odd), (525.0, 525.0)) and o.principal_point == (31.0, 23.0)
    o2 = ImageDepthSource()
    assert o2.initialize_directory(str(odd), (525.0, 525.0), (0.0, 20.0)) and o2.principal_point == (31.0, 23.0)
    o3 = ImageDepthSource()
    assert o3.initialize_directory(str(odd), (525.0, 525.0), (30.0, 0.0)) and o3.principal_point == (30.0, 0.0)
    g = ImageDepthSource()
    with pytest.raises(ValueError):
        g.initialize_directory(str(tmp_path), (525.0, 525.0), depth_width=32, depth_height=48)


@pytest.mark.gpu
def test_vertex_and_normal_maps_from_png(gpu, tmp_path):
    """depthToVertices (organized_point_cloud.cu:11-40): ((u-pp.x) z/f.x, (v-pp.y) z/f.y, z, 1) for z in range, else w = 0"""
    frames = write_sequence(tmp_path, n=1, w=80, h=60, seed=3)
    s = ImageDepthSource()
    assert s.initialize_directory(str(tmp_path), (100.0, 110.0), (40.0, 30.0), scale_to_meters=0.001)
    depth, vert, norm = s.vertices_and_normals(depth_range=(0.4, 1.4))
    z = frames[0].astype(np.float32) * np.float32(0.001)
    assert np.array_equal(depth.cpu().numpy(), z)
    v = vert.cpu().numpy()
    u, w_ = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    ok = (z >= 0.4) & (z <= 1.4)
    assert np.array_equal(v[..., 3] > 0, ok)
    np.testing.assert_allclose(v[..., 0][ok], ((u - 40.0) * z / 100.0)[ok], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(v[..., 1][ok], ((w_ - 30.0) * z / 110.0)[ok], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(v[..., 2][ok], z[ok])
    n = norm.cpu().numpy()
    assert n.shape == v.shape and np.isfinite(n).all()
