"""Host-side model: kinematic tree, geometry, voxelisation into per-frame SDFs (the cold path that feeds the
device model). Mirrors the behaviour of the reference's HostOnlyModel + readModelXML:

  * frames are numbered depth-first in document order, joint i drives frame i+1 (dart_io.cpp:15-117,
    host_only_model.cpp:58-142, mirrored_model.h:38);
  * dependency mask row = frame, 1 iff the joint is on the chain root->frame (host_only_model.cpp:97-103);
  * model version 1 orientations are ZYX Euler angles passed as (z,y,x) (host_only_model.cpp:134-136,196);
  * voxelize(): occupancy of the frame's primitives (voxelizeFrame, host_only_model.cpp:530-831, including its
    quirks: unrotated cylinders/cubes ignore the geometry translation, rotated ones apply the rotation only)
    followed by the signed distance transform semantics of distance_transforms.cu:540-562 (0 on the inside
    boundary layer, voxel units);
  * voxelize2(): analytic sphere SDF (sdf.cpp:204-225), used by the reference's own kernel test.

Everything here is numpy on the host; nothing here runs per Gauss-Newton iteration.
"""
import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import se3

F = np.float32
ROTATIONAL, PRISMATIC = 0, 1
GEOM_SPHERE, GEOM_CYLINDER, GEOM_CUBE, GEOM_MESH = 0, 1, 2, 4
_GEOM_NAMES = {"sphere": GEOM_SPHERE, "cylinder": GEOM_CYLINDER, "cube": GEOM_CUBE, "mesh": GEOM_MESH}


class Grid:
    """Grid3D<float>: dim (x,y,z), offset, resolution, data[z][y][x] in voxel units (grid_3d.h:423-426)."""

    def __init__(self, dim, offset, resolution, data=None):
        self.dim = tuple(int(v) for v in dim)
        self.offset = np.asarray(offset, dtype=F)
        self.resolution = F(resolution)
        if data is None:
            data = np.zeros((self.dim[2], self.dim[1], self.dim[0]), dtype=F)
        self.data = np.ascontiguousarray(data, dtype=F).reshape(self.dim[2], self.dim[1], self.dim[0])

    def voxel_centers_world(self):
        xs = (np.arange(self.dim[0], dtype=F) + F(0.5)) * self.resolution + self.offset[0]
        ys = (np.arange(self.dim[1], dtype=F) + F(0.5)) * self.resolution + self.offset[1]
        zs = (np.arange(self.dim[2], dtype=F) + F(0.5)) * self.resolution + self.offset[2]
        return xs, ys, zs


def _safe_eval(expr, params):
    """Evaluate a libmatheval-style size expression (host_only_model.cpp:37-56) over the <param> table."""
    if isinstance(expr, (int, float)):
        return float(expr)
    expr = str(expr).strip()
    try:
        return float(expr)
    except ValueError:
        pass
    allowed = {k: float(v) for k, v in params.items()}
    allowed.update({"sqrt": math.sqrt, "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp,
                    "log": math.log, "abs": abs, "pi": math.pi})
    code = compile(expr.replace("^", "**"), "<expr>", "eval")
    for name in code.co_names:
        if name not in allowed:
            raise ValueError("unknown symbol %r in model expression %r" % (name, expr))
    return float(eval(code, {"__builtins__": {}}, allowed))


class HostModel:
    def __init__(self, version=1, name="model"):
        self.version = version
        self.name = name
        self.params = {}
        self.frame_parent = [-1]
        self.joint_type = []
        self.joint_axis = []
        self.joint_limits = []
        self.joint_names = []
        self.joint_position = []
        self.joint_orientation = []
        self.dh = []
        self.T_pf = []
        self.dependencies = [[]]          # per frame, list over joints
        self.geoms = []                   # dicts: frame, type, scale, T(12), mesh
        self.sdfs = []                    # list of Grid
        self.sdf_frame = []
        self.frame_sdf = []

    # ---- structure -----------------------------------------------------------------------------------
    @property
    def n_frames(self):
        return len(self.frame_parent)

    @property
    def n_joints(self):
        return len(self.joint_type)

    @property
    def dims(self):
        return 6 + self.n_joints

    def add_frame(self, parent, joint_type, pos, ori, axis, jmin, jmax, name="", dh=(0, 0, 0, 0)):
        pos = [F(_safe_eval(v, self.params)) for v in pos]
        ori = [F(_safe_eval(v, self.params)) for v in ori]
        axis = [F(_safe_eval(v, self.params)) for v in axis]
        dh = [F(_safe_eval(v, self.params)) for v in dh]
        frame = self.n_frames
        self.frame_parent.append(int(parent))
        self.joint_type.append(int(joint_type))
        self.joint_limits.append((F(_safe_eval(jmin, self.params)), F(_safe_eval(jmax, self.params))))
        self.joint_names.append(name)
        self.joint_position.append(pos)
        self.joint_orientation.append(ori)
        self.joint_axis.append(axis)
        self.dh.append(dh)
        for row in self.dependencies:
            row.append(0)
        self.dependencies.append(list(self.dependencies[parent][:-1]) + [1])
        if self.version == 0:
            T = se3.mul(se3.exp([pos[0], pos[1], pos[2], 0, 0, 0]), se3.exp([0, 0, 0, ori[0], ori[1], ori[2]]))
        else:
            T = se3.mul(se3.translation(*pos), se3.from_euler(ori[2], ori[1], ori[0]))
        self.T_pf.append(T)
        return frame

    def add_geometry(self, frame, gtype, scale, trans, rot, mesh=None):
        scale = [F(_safe_eval(v, self.params)) for v in scale]
        trans = [F(_safe_eval(v, self.params)) for v in trans]
        rot = [F(_safe_eval(v, self.params)) for v in rot]
        if self.version == 0:
            T = se3.exp([trans[0], trans[1], trans[2], rot[0], rot[1], rot[2]])
        else:
            T = se3.mul(se3.translation(*trans), se3.from_euler(rot[2], rot[1], rot[0]))
        self.geoms.append({"frame": int(frame), "type": int(gtype), "scale": np.asarray(scale, dtype=F), "T": T, "mesh": mesh})

    def frame_geoms(self, frame):
        return [g for g in self.geoms if g["frame"] == frame]

    # ---- voxelisation --------------------------------------------------------------------------------
    def _frame_extent(self, geoms):
        mn = np.full(3, np.inf, dtype=F)
        mx = np.full(3, -np.inf, dtype=F)
        for g in geoms:
            s = g["scale"]
            t6 = se3.log(g["T"])
            c, w = t6[:3], t6[3:]
            rotated = bool(np.any(w != 0))
            R = se3.exp([0, 0, 0, w[0], w[1], w[2]]).reshape(3, 4)[:, :3].astype(np.float64)
            if g["type"] == GEOM_SPHERE:
                if not rotated:
                    o, sz = c - s, 2 * s
                else:   # aabbEllipsoid, geometry.cpp:614-630
                    d = np.sqrt(((s.astype(np.float64) ** 2)[None, :] * R ** 2).sum(axis=1))
                    o, sz = c - d, 2 * d
            elif g["type"] == GEOM_CYLINDER:
                if not rotated:
                    o = np.array([c[0] - s[0], c[1] - s[1], c[2]])
                    sz = np.array([2 * s[0], 2 * s[1], s[2]])
                else:   # aabbEllipticCylinder, geometry.cpp:633-658
                    u, v, c2 = R @ [s[0], 0, 0], R @ [0, s[1], 0], R @ [0, 0, s[2]]
                    r = np.sqrt(u ** 2 + v ** 2)
                    o = np.minimum(c - r, c + c2 - r)
                    sz = np.maximum(c + r, c + c2 + r) - o
            elif g["type"] == GEOM_CUBE:
                if not rotated:
                    o, sz = c - 0.5 * s, s
                else:   # aabbRectangularPrism, geometry.cpp:661-691 (as written there: extents only)
                    corners = np.array([[sx * s[0], sy * s[1], szz * s[2]] for sx in (-1, 1) for sy in (-1, 1) for szz in (-1, 1)])
                    corners = corners @ R.T
                    o, sz = corners.min(axis=0), corners.max(axis=0)
            elif g["type"] == GEOM_MESH:    # bounds of the scaled, transformed vertices (host_only_model.cpp:591-628)
                lo, hi = self._geom_mesh(g).bounds()
                o, sz = lo, (hi - lo).astype(F)
            else:
                raise NotImplementedError("geometry type %r" % (g["type"],))
            mn = np.minimum(mn, np.asarray(o, dtype=F))
            mx = np.maximum(mx, np.asarray(o, dtype=F) + np.asarray(sz, dtype=F))
        return mn, mx

    @staticmethod
    def _geom_mesh(g):
        """scaleMesh + transformMesh of a mesh geometry (host_only_model.cpp:593-601, 815-824)."""
        if g.get("mesh") is None:
            raise ValueError("mesh geometry without a loaded mesh (read_model_xml(..., load_meshes=True))")
        return g["mesh"].scaled(g["scale"]).transformed(g["T"])

    def _occupancy(self, frame, resolution, padding):
        geoms = self.frame_geoms(frame)
        mn, mx = self._frame_extent(geoms)
        mn = (mn - F(padding)).astype(F)
        mx = (mx + F(padding)).astype(F)
        dim = np.ceil((mx - mn) / F(resolution)).astype(np.int64)
        grid = Grid(dim, mn, resolution)
        occ = np.zeros((dim[2], dim[1], dim[0]), dtype=bool)
        res = F(resolution)
        for g in geoms:
            if g["type"] == GEOM_MESH:      # splatSolidMesh writes the foreground value 0 (host_only_model.cpp:813-826)
                from .mesh import splat_solid_mesh
                data = splat_solid_mesh(self._geom_mesh(g), dim, mn, res, np.where(occ, F(0), F(1e20)).astype(F))
                occ = data == 0
                continue
            s = g["scale"].astype(F)
            t6 = se3.log(g["T"])
            c, w = t6[:3], t6[3:]
            rotated = bool(np.any(w != 0))
            off = (mn + F(0.5) * res).astype(F)
            vx = lambda o: (np.arange(dim[0], dtype=F) * res + o).astype(F)   # noqa: E731
            vy = lambda o: (np.arange(dim[1], dtype=F) * res + o).astype(F)   # noqa: E731
            vz = lambda o: (np.arange(dim[2], dtype=F) * res + o).astype(F)   # noqa: E731
            if g["type"] == GEOM_SPHERE and not rotated:
                X, Y, Z = vx(off[0] - c[0]), vy(off[1] - c[1]), vz(off[2] - c[2])
                A = (X * X / (s[0] * s[0]))[None, None, :]
                B = (Y * Y / (s[1] * s[1]))[None, :, None]
                C = (Z * Z / (s[2] * s[2]))[:, None, None]
                occ |= np.sqrt(A + B + C) < 1.0
                continue
            if g["type"] == GEOM_CYLINDER and not rotated:
                X, Y, Z = vx(off[0]), vy(off[1]), vz(off[2])      # translation ignored (host_only_model.cpp:729-746)
                A = (Y * Y / (s[1] * s[1]))[None, :, None]
                B = (X * X / (s[0] * s[0]))[None, None, :]
                inz = ((Z >= 0) & (Z < s[2]))[:, None, None]
                occ |= (np.sqrt(A + B) < 1.0) & inz
                continue
            if g["type"] == GEOM_CUBE and not rotated:
                X, Y, Z = vx(off[0]), vy(off[1]), vz(off[2])      # translation ignored (:773-790)
                inx = ~((X > 0.5 * s[0]) | (X < -0.5 * s[0]))
                iny = ~((Y > 0.5 * s[1]) | (Y < -0.5 * s[1]))
                inz = ~((Z > 0.5 * s[2]) | (Z < -0.5 * s[2]))
                occ |= inz[:, None, None] & iny[None, :, None] & inx[None, None, :]
                continue
            # rotated primitives: voxel centre -> geometry frame
            X, Y, Z = np.meshgrid(vx(off[0]), vy(off[1]), vz(off[2]), indexing="xy")
            P = np.stack([X, Y, Z], axis=-1).transpose(2, 0, 1, 3)     # [z][y][x][3]
            if g["type"] == GEOM_SPHERE:
                Tinv = se3.invert(se3.exp([c[0], c[1], c[2], w[0], w[1], w[2]])).reshape(3, 4)
                Q = P @ Tinv[:, :3].T + Tinv[:, 3]
                occ |= np.sqrt((Q[..., 0] / s[0]) ** 2 + (Q[..., 1] / s[1]) ** 2 + (Q[..., 2] / s[2]) ** 2) < 1.0
            else:
                Rinv = se3.invert(se3.exp([0, 0, 0, w[0], w[1], w[2]])).reshape(3, 4)   # rotation only (:748-750,:792-794)
                Q = P @ Rinv[:, :3].T
                if g["type"] == GEOM_CYLINDER:
                    occ |= (np.sqrt((Q[..., 0] / s[0]) ** 2 + (Q[..., 1] / s[1]) ** 2) < 1.0) & (Q[..., 2] >= 0) & (Q[..., 2] < s[2])
                else:
                    occ |= ((Q[..., 0] > -0.5 * s[0]) & (Q[..., 0] < 0.5 * s[0]) & (Q[..., 1] > -0.5 * s[1]) &
                            (Q[..., 1] < 0.5 * s[1]) & (Q[..., 2] > -0.5 * s[2]) & (Q[..., 2] < 0.5 * s[2]))
        return grid, occ

    @staticmethod
    def signed_distance_from_occupancy(occ):
        """signedDistanceTransform3D<float,true> semantics (distance_transforms.cu:173-263, 540-562): outside =
        Euclidean distance (voxels) to the nearest inside voxel; inside = -(distance to the nearest voxel that
        is outside or lies on the inside boundary layer), hence 0 on that layer."""
        from scipy import ndimage
        if not occ.any():
            return np.full(occ.shape, F(1e10), dtype=F)
        outside = ndimage.distance_transform_edt(~occ)
        pad = np.pad(occ, 1, mode="constant", constant_values=True)   # out-of-grid neighbours never count as outside
        nb_out = (~pad[:-2, 1:-1, 1:-1]) | (~pad[2:, 1:-1, 1:-1]) | (~pad[1:-1, :-2, 1:-1]) | (~pad[1:-1, 2:, 1:-1]) | \
                 (~pad[1:-1, 1:-1, :-2]) | (~pad[1:-1, 1:-1, 2:])
        seed_zero = (~occ) | (occ & nb_out)
        inside = ndimage.distance_transform_edt(~seed_zero)
        return np.where(occ, -inside, outside).astype(F)

    def _reset_sdfs(self):
        self.sdfs, self.sdf_frame, self.frame_sdf = [], [], []

    def voxelize(self, resolution=0.005, padding=0.10):
        """HostOnlyModel::voxelize, host_only_model.cpp:259-335."""
        self._reset_sdfs()
        for f in range(self.n_frames):
            if not self.frame_geoms(f):
                self.frame_sdf.append(-1)
                continue
            grid, occ = self._occupancy(f, resolution, padding)
            grid.data = self.signed_distance_from_occupancy(occ)
            self.frame_sdf.append(len(self.sdfs))
            self.sdfs.append(grid)
            self.sdf_frame.append(f)
        return self

    def voxelize2(self, resolution=0.005, padding=0.10):
        """HostOnlyModel::voxelize2, host_only_model.cpp:337-432: analytic SDFs (spheres only here; the
        reference's cube case falls through into the sphere case and it has no cylinder case, Appendix A-12)."""
        self._reset_sdfs()
        for f in range(self.n_frames):
            geoms = self.frame_geoms(f)
            if not geoms:
                self.frame_sdf.append(-1)
                continue
            grid, _ = self._occupancy(f, resolution, padding)
            data = np.full(grid.data.shape, F(1e20), dtype=F)
            xs, ys, zs = grid.voxel_centers_world()
            for g in geoms:
                if g["type"] != GEOM_SPHERE:
                    raise NotImplementedError("voxelize2 supports sphere geometry only")
                T = g["T"].reshape(3, 4)     # analyticSphereSdf applies the geometry transform to the centre
                X = xs[None, None, :].astype(F); Y = ys[None, :, None].astype(F); Z = zs[:, None, None].astype(F)
                cx = (T[0, 0] * X + T[0, 1] * Y + T[0, 2] * Z + T[0, 3]).astype(F)
                cy = (T[1, 0] * X + T[1, 1] * Y + T[1, 2] * Z + T[1, 3]).astype(F)
                cz = (T[2, 0] * X + T[2, 1] * Y + T[2, 2] * Z + T[2, 3]).astype(F)
                dist = ((np.sqrt((cx * cx + cy * cy + cz * cz).astype(F)).astype(F) - g["scale"][0]) / grid.resolution).astype(F)
                data = np.where(np.abs(dist) < np.abs(data), dist, data)
            grid.data = data
            self.frame_sdf.append(len(self.sdfs))
            self.sdfs.append(grid)
            self.sdf_frame.append(f)
        return self

    # ---- flat arrays for the C ABI ---------------------------------------------------------------------
    def arrays(self):
        nJ = self.n_joints
        a = {
            "frame_parent": np.asarray(self.frame_parent, dtype=np.int32),
            "joint_type": np.asarray(self.joint_type, dtype=np.int32).reshape(nJ),
            "joint_axis": np.asarray(self.joint_axis, dtype=F).reshape(nJ, 3),
            "T_pf": np.asarray(self.T_pf, dtype=F).reshape(nJ, 12),
            "dh": np.asarray(self.dh, dtype=F).reshape(nJ, 4),
            "joint_limits": np.asarray(self.joint_limits, dtype=F).reshape(nJ, 2),
            "sdf_frame": np.asarray(self.sdf_frame, dtype=np.int32),
            "dependencies": np.asarray([row + [0] * (nJ - len(row)) for row in self.dependencies], dtype=np.int32).reshape(self.n_frames, nJ),
            "sdf_dim": np.asarray([g.dim for g in self.sdfs], dtype=np.int32).reshape(-1, 3),
            "sdf_offset": np.asarray([g.offset for g in self.sdfs], dtype=F).reshape(-1, 3),
            "sdf_resolution": np.asarray([g.resolution for g in self.sdfs], dtype=F),
            "sdf_data": [np.ascontiguousarray(g.data, dtype=F) for g in self.sdfs],
        }
        return a

    # ---- host FK (cold path: bounding boxes, scene set-up) ------------------------------------------------
    def forward_kinematics(self, joints, mode="dh"):
        """T_mf per frame. mode 'dh' = MirroredModel::setPose (mirrored_model.cpp:185-225), 'axis' =
        setArticulation / HostOnlyModel::setPose (axis-angle)."""
        T_mf = [se3.identity()]
        for f in range(1, self.n_frames):
            j = f - 1
            lo, hi = self.joint_limits[j]
            p = F(min(max(lo, F(joints[j])), hi))
            ax = self.joint_axis[j]
            if mode == "dh":
                a, alpha, d, theta = self.dh[j]
                if self.joint_type[j] == ROTATIONAL:
                    D = se3.from_dh_graspit(theta + p * ax[2], d, a, alpha + p * ax[0], p * ax[1])
                else:
                    D = se3.from_dh_graspit(theta, d + p * ax[2], a + p * ax[0], alpha, p * ax[1])
            else:
                if self.joint_type[j] == ROTATIONAL:
                    D = se3.exp([0, 0, 0, p * ax[0], p * ax[1], p * ax[2]])
                else:
                    D = se3.exp([p * ax[0], p * ax[1], p * ax[2], 0, 0, 0])
            T = se3.mul(self.T_pf[j], D)
            T_mf.append(se3.mul(T_mf[self.frame_parent[f]], T))
        return np.asarray(T_mf, dtype=F)

    def articulated_bounding_box(self, padding, n_sweep=4):
        """Model::getArticulatedBoundingBox, model.cpp:495-569."""
        g0 = self.sdfs[0]
        mins = g0.offset + F(padding)
        maxs = g0.offset - F(padding) + g0.resolution * np.asarray(g0.dim, dtype=F)
        dep = self.arrays()["dependencies"]
        for s in range(1, len(self.sdfs)):
            g = self.sdfs[s]
            frame = self.sdf_frame[s]
            deps = [j for j in range(self.n_joints) if dep[frame, j]]
            if not deps:
                continue
            lo = g.offset + F(padding)
            hi = g.offset - F(padding) + g.resolution * np.asarray(g.dim, dtype=F)
            corners = np.array([[lo[0] if i & 4 else hi[0], lo[1] if i & 2 else hi[1], lo[2] if i & 1 else hi[2]] for i in range(8)], dtype=F)
            sweeps = [np.linspace(self.joint_limits[j][0], self.joint_limits[j][1], n_sweep, dtype=F) for j in deps]
            for combo in np.array(np.meshgrid(*sweeps, indexing="ij")).reshape(len(deps), -1).T:
                joints = np.zeros(self.n_joints, dtype=F)
                joints[deps] = combo
                T = self.forward_kinematics(joints, mode="axis")[frame]
                cm = se3.apply(T, corners)
                mins = np.minimum(mins, cm.min(axis=0))
                maxs = np.maximum(maxs, cm.max(axis=0))
        return mins.astype(F), maxs.astype(F)

    def obs_sdf_layout(self, padding=0.10, obs_sdf_size=64, obs_padding=0.02):
        """Observation-SDF sizing of Tracker::addModel, tracker.cpp:92-115."""
        if self.dims == 6:
            g = self.sdfs[0]
            size_m = g.resolution * np.asarray(g.dim, dtype=F)
            center = g.offset + F(0.5) * size_m
            obs_size = size_m - 2 * F(padding) + 2 * F(obs_padding)
            max_dim = F(obs_size.max())
            return (obs_sdf_size,) * 3, (center - max_dim / 2).astype(F), F(max_dim / obs_sdf_size)
        mins, maxs = self.articulated_bounding_box(padding)
        obs_size = maxs - mins + 2 * F(obs_padding)
        max_dim = F(obs_size.max())
        offset = mins + F(0.5) * (obs_size - max_dim) - F(obs_padding)
        return (obs_sdf_size,) * 3, offset.astype(F), F(max_dim / obs_sdf_size)

    # ---- collision cloud (tracker.cpp:143-210; mesh_sample.cpp:13-72 samples area-weighted with rand()) ------
    def collision_cloud(self, density=1e5, seed=0):
        """Surface samples per geometry in the owning frame's coordinates, .w = SDF number. The reference draws
        from unseeded rand() (so its cloud changes run to run, Appendix A-18); here a seeded generator samples
        the analytic primitive surfaces at the same density. The samples of one SDF are stored along a Z-order
        curve (spatial.py: consecutive points are neighbours, which is what the device kernels' early exits need)."""
        from .spatial import morton_order
        rng = np.random.default_rng(seed)
        pts = []
        for f in range(self.n_frames):
            s_num = self.frame_sdf[f] if self.frame_sdf else -1
            if s_num < 0:
                continue
            first = len(pts)
            for g in self.frame_geoms(f):
                sc = g["scale"].astype(np.float64)
                if g["type"] == GEOM_SPHERE:
                    # Knud Thomsen's ellipsoid-area approximation
                    p = 1.6075
                    area = 4 * math.pi * (((sc[0] * sc[1]) ** p + (sc[0] * sc[2]) ** p + (sc[1] * sc[2]) ** p) / 3) ** (1 / p)
                    n = int(round(area * density))
                    v = rng.normal(size=(n, 3))
                    v /= np.linalg.norm(v, axis=1, keepdims=True)
                    local = v * sc
                elif g["type"] == GEOM_CYLINDER:
                    side = 2 * math.pi * 0.5 * (sc[0] + sc[1]) * sc[2]
                    cap = math.pi * sc[0] * sc[1]
                    n = int(round((side + 2 * cap) * density))
                    u = rng.random(n)
                    th = rng.random(n) * 2 * math.pi
                    on_side = u < side / (side + 2 * cap)
                    r = np.where(on_side, 1.0, np.sqrt(rng.random(n)))
                    z = np.where(on_side, rng.random(n) * sc[2], np.where(rng.random(n) < 0.5, 0.0, sc[2]))
                    local = np.stack([r * np.cos(th) * sc[0], r * np.sin(th) * sc[1], z], axis=1)
                elif g["type"] == GEOM_CUBE:
                    areas = np.array([sc[1] * sc[2], sc[0] * sc[2], sc[0] * sc[1]]) * 2
                    n = int(round(areas.sum() * density))
                    face = rng.choice(3, size=n, p=areas / areas.sum())
                    local = (rng.random((n, 3)) - 0.5) * sc
                    sign = np.where(rng.random(n) < 0.5, -0.5, 0.5)
                    local[np.arange(n), face] = sign * sc[face]
                elif g["type"] == GEOM_MESH:    # sampleMesh on the mesh as placed in the frame (tracker.cpp:143-210)
                    from .mesh import sample_mesh, GlibcRand
                    world = sample_mesh(self._geom_mesh(g), density, GlibcRand(1 + seed))
                    pts.append(np.concatenate([world.astype(F), np.full((len(world), 1), s_num, dtype=F)], axis=1))
                    continue
                else:
                    continue
                world = se3.apply(g["T"], local.astype(F))
                pts.append(np.concatenate([world.astype(F), np.full((len(world), 1), s_num, dtype=F)], axis=1))
            if len(pts) > first:
                grp = np.concatenate(pts[first:], axis=0)
                pts[first:] = [grp[morton_order(grp[:, :3])]]
        if not pts:
            return np.zeros((0, 4), dtype=F)
        return np.ascontiguousarray(np.concatenate(pts, axis=0), dtype=F)

    # ---- (de)serialisation -------------------------------------------------------------------------------
    def to_json(self):
        frames = []
        for j in range(self.n_joints):
            frames.append({"parent": self.frame_parent[j + 1], "type": self.joint_type[j], "name": self.joint_names[j],
                           "position": [float(v) for v in self.joint_position[j]],
                           "orientation": [float(v) for v in self.joint_orientation[j]],
                           "axis": [float(v) for v in self.joint_axis[j]],
                           "limits": [float(v) for v in self.joint_limits[j]], "dh": [float(v) for v in self.dh[j]]})
        geoms = []
        for g in self._geom_specs:
            spec = {"frame": g["frame"], "type": g["type"], "scale": [float(v) for v in g["scale"]], "trans": g["trans"], "rot": g["rot"]}
            if g.get("meshFile"):
                spec["meshFile"] = g["meshFile"]       # from_json(spec, mesh_dir=...) reads it again
            geoms.append(spec)
        return {"name": self.name, "version": self.version, "frames": frames, "geoms": geoms}

    _geom_specs = ()

    @classmethod
    def from_json(cls, spec, mesh_dir=None):
        if isinstance(spec, str):
            if mesh_dir is None:
                mesh_dir = os.path.dirname(os.path.abspath(spec))
            with open(spec) as fh:
                spec = json.load(fh)
        m = cls(version=spec.get("version", 1), name=spec.get("name", "model"))
        for fr in spec["frames"]:
            m.add_frame(fr["parent"], fr["type"], fr["position"], fr["orientation"], fr["axis"], fr["limits"][0],
                        fr["limits"][1], fr.get("name", ""), fr.get("dh", (0, 0, 0, 0)))
        m._geom_specs = list(spec.get("geoms", []))
        for g in m._geom_specs:
            mesh = None
            if g["type"] == GEOM_MESH:
                if not g.get("meshFile") or mesh_dir is None:
                    raise ValueError("mesh geometry in a model description needs meshFile and a mesh directory")
                from .mesh import read_mesh
                mesh = read_mesh(os.path.join(mesh_dir, g["meshFile"]))
            m.add_geometry(g["frame"], g["type"], g["scale"], g["trans"], g["rot"], mesh=mesh)
        return m


def read_model_xml(path, mesh_proxy=None, load_meshes=False):
    """readModelXML + readFrameXML (dart_io.cpp:15-159). Mesh geoms: with `load_meshes` the file named by `meshFile`
    (relative to the XML's directory) is read by dart_amd.mesh.read_mesh and voxelised / sampled by the mesh pipeline;
    otherwise `mesh_proxy(frame_index, geom_attrib_dict, model)` may return a list of primitive geometry specs
    ({type, scale, trans, rot}) that stand in for a mesh that is not available, or None to drop it."""
    root = ET.parse(path).getroot()
    if root.tag != "model":
        raise ValueError("expected 'model' at root node; got %r" % root.tag)
    m = HostModel(version=int(root.attrib.get("version", 0)), name=os.path.splitext(os.path.basename(path))[0])
    for child in root:
        if child.tag == "param":
            m.params[child.attrib["name"]] = float(child.attrib["value"])
    specs = []

    def walk(parent, element):
        for child in element:
            if child.tag == "frame":
                a = child.attrib
                jt = {"rotational": ROTATIONAL, "prismatic": PRISMATIC}[a["jointType"]]
                pos = ori = axis = ("0", "0", "0")
                dh = ("0", "0", "0", "0")
                for gc in child:
                    if gc.tag == "position":
                        pos = (gc.attrib["x"], gc.attrib["y"], gc.attrib["z"])
                    elif gc.tag == "orientation":
                        ori = (gc.attrib["x"], gc.attrib["y"], gc.attrib["z"])
                    elif gc.tag == "axis":
                        axis = (gc.attrib["x"], gc.attrib["y"], gc.attrib["z"])
                    elif gc.tag == "dh_offset":
                        dh = (gc.attrib["a"], gc.attrib["alpha"], gc.attrib["d"], gc.attrib["theta"])
                fid = m.add_frame(parent, jt, pos, ori, axis, a["jointMin"], a["jointMax"], a.get("jointName", ""), dh)
                walk(fid, child)
            elif child.tag == "geom":
                a = child.attrib
                gt = _GEOM_NAMES[a["type"]]
                spec = {"frame": parent, "type": gt, "scale": [a["sx"], a["sy"], a["sz"]],
                        "trans": [a["tx"], a["ty"], a["tz"]], "rot": [a["rx"], a["ry"], a["rz"]]}
                if gt == GEOM_MESH and load_meshes:
                    from .mesh import read_mesh
                    spec["mesh"] = read_mesh(os.path.join(os.path.dirname(os.path.abspath(path)), a["meshFile"]))
                    spec["meshFile"] = a["meshFile"]
                    specs.append(spec)
                elif gt == GEOM_MESH:
                    repl = mesh_proxy(parent, dict(a), m) if mesh_proxy else None
                    for r in (repl or []):
                        r = dict(r); r["frame"] = parent
                        specs.append(r)
                else:
                    specs.append(spec)

    walk(0, root)
    norm = []
    for s in specs:
        s = dict(s)
        s["scale"] = [_safe_eval(v, m.params) for v in s["scale"]]
        s["trans"] = [_safe_eval(v, m.params) for v in s["trans"]]
        s["rot"] = [_safe_eval(v, m.params) for v in s["rot"]]
        mesh = s.pop("mesh", None)
        norm.append(s)
        m.add_geometry(s["frame"], s["type"], s["scale"], s["trans"], s["rot"], mesh=mesh)
    m._geom_specs = norm
    return m


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_asset(name):
    """Load one of the bundled kinematic descriptions (dart_amd/assets/<name>.json, written by
    tools/convert_models.py from the reference's model XMLs; mesh links replaced by declared primitive proxies)."""
    return HostModel.from_json(os.path.join(ASSET_DIR, name + ".json"))


def sphere_model(radius=0.1):
    """The rigid single-sphere model of the reference's kernel tests (test_obs_to_mod_kernels.cpp:18-25)."""
    m = HostModel(version=0, name="sphere")
    m._geom_specs = [{"frame": 0, "type": GEOM_SPHERE, "scale": [radius] * 3, "trans": [0, 0, 0], "rot": [0, 0, 0]}]
    m.add_geometry(0, GEOM_SPHERE, [radius] * 3, [0, 0, 0], [0, 0, 0])
    return m
