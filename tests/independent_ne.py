"""An independent fp64 construction of the obs->mod normal equations, used to pin BOTH the CPU oracle's and the HIP path's
packed systems without going through either one's packing code.

Built from the reference source directly (obsToMod.cu:246-337, kernel_common.h:141-153, grid_3d.h:86-112,314-330):
  * the SDF value and gradient are re-derived here in numpy fp64 from the raw voxel arrays (trilinear interpolation and the
    six-lookup central difference), not taken from the oracle;
  * the per-point Jacobian row comes from orc_error_jacobian, which the reference's own finite-difference test pins (KAT 3);
  * packing, residual sign and the Huber branch are restated from obsToMod.cu:301-335.
Test infrastructure only."""
import numpy as np

import oracle_lib as O

F = np.float32


def trilinear64(grid, pg):
    """getValueInterpolated (grid_3d.h:86-112) in fp64. grid: dart_amd HostModel SDF (data [z,y,x], voxel units); pg [N,3]."""
    t = np.asarray(pg, dtype=np.float64) - 0.5
    i0 = np.floor(t).astype(np.int64)
    f = t - i0
    d = np.asarray(grid.data, dtype=np.float64)

    def v(dx, dy, dz):
        return d[i0[:, 2] + dz, i0[:, 1] + dy, i0[:, 0] + dx]
    dx00 = v(0, 0, 0) + f[:, 0] * (v(1, 0, 0) - v(0, 0, 0))
    dx01 = v(0, 0, 1) + f[:, 0] * (v(1, 0, 1) - v(0, 0, 1))
    dx10 = v(0, 1, 0) + f[:, 0] * (v(1, 1, 0) - v(0, 1, 0))
    dx11 = v(0, 1, 1) + f[:, 0] * (v(1, 1, 1) - v(0, 1, 1))
    dxy0 = dx00 + f[:, 1] * (dx10 - dx00)
    dxy1 = dx01 + f[:, 1] * (dx11 - dx01)
    return dxy0 + f[:, 2] * (dxy1 - dxy0)


def gradient64(grid, pg):
    """getGradientInterpolated (grid_3d.h:314-330): 0.5 * (f(p + e_k) - f(p - e_k)), voxel units."""
    pg = np.asarray(pg, dtype=np.float64)
    out = np.zeros_like(pg)
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1.0
        out[:, k] = 0.5 * (trilinear64(grid, pg + e) - trilinear64(grid, pg - e))
    return out


def apply64(T, p):
    T = np.asarray(T, dtype=np.float64).reshape(3, 4)
    return p @ T[:, :3].T + T[:, 3]


def rotate64(T, p):
    T = np.asarray(T, dtype=np.float64).reshape(3, 4)
    return p @ T[:, :3].T


def pack(JTe, JTJ, e):
    d = len(JTe)
    out = np.zeros(d + d * (d + 1) // 2 + 1)
    out[:d] = JTe
    for i in range(d):
        for j in range(i + 1):
            out[d + i * (i + 1) // 2 + j] = JTJ[i, j]
    out[-1] = e
    return out


def obs2mod_system(host_model, om, vert, T_mc, T_fm, T_mf, pts, huber_delta=0.02, residual_override=None):
    """Packed system [J^T e | lower-tri J^T J | e] (kernel_common.h:141-153 layout) in fp64 for the associated points `pts`
    (index, dataAssociation). Returns (packed, errors) where errors are the independently interpolated SDF distances in
    metres. residual_override: per-point residuals to use instead (e.g. the analytic sphere distance)."""
    d = om.dims
    JTe, JTJ, e = np.zeros(d), np.zeros((d, d)), 0.0
    p = vert.reshape(-1, 4)
    errs = np.zeros(len(pts))
    for k in range(len(pts)):
        xc = p[pts["index"][k], :3].astype(np.float64)
        g = int(pts["dataAssociation"][k])
        sdf = host_model.sdfs[g]
        frame = int(host_model.sdf_frame[g])
        xm = apply64(T_mc, xc[None])[0]
        xf = apply64(T_fm[frame], xm[None])[0]
        xg = (xf - np.asarray(sdf.offset, dtype=np.float64)) / float(sdf.resolution)
        err = float(trilinear64(sdf, xg[None])[0]) * float(sdf.resolution)
        errs[k] = err
        if residual_override is not None:
            err = float(residual_override[k])
        gf = gradient64(sdf, xg[None])[0]
        gm = rotate64(T_mf[frame], gf[None])[0]
        J = O.error_jacobian(om, T_fm, T_mf, xm.astype(F), frame, gm.astype(F)).astype(np.float64)
        if abs(err) < huber_delta:                      # computeSquaredLossResult(dims, -residual, ...), obsToMod.cu:315-316
            r = -err
            JTe += J * r
            JTJ += np.outer(J, J)
            e += 0.5 * r * r
        else:                                           # obsToMod.cu:318-332: +-delta gradient, unweighted J^T J
            v = -huber_delta if err < 0 else huber_delta
            JTe += v * -J
            JTJ += np.outer(J, J)
            e += huber_delta * (abs(err) - 0.5 * huber_delta)
    return pack(JTe, JTJ, e), errs


def rel_err_blocks(got, ref, dims):
    """max over {J^T e, J^T J, e} of ||got-ref|| / ||ref|| (norm-relative per block)."""
    out = 0.0
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    for sl in (slice(0, dims), slice(dims, len(ref) - 1), slice(len(ref) - 1, len(ref))):
        n = np.linalg.norm(ref[sl])
        dd = np.linalg.norm(got[sl] - ref[sl])
        out = max(out, dd / n if n > 0 else dd)
    return out
