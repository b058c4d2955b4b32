"""Host-side SE3 helpers (numpy float32, row-major 3x4 as float[12]) used for scene set-up and model loading.

These mirror the reference's host-callable constructors (src/geometry/SE3.h) closely enough for cold-path use;
the per-iteration arithmetic lives in the HIP library.
"""
import numpy as np

F = np.float32


def identity():
    return np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=F)


def to44(T):
    M = np.eye(4, dtype=np.float64)
    M[:3, :] = np.asarray(T, dtype=np.float64).reshape(3, 4)
    return M


def from44(M):
    return np.asarray(M, dtype=np.float64)[:3, :].astype(F).reshape(12)


def mul(A, B):
    """SE3Transform(A,B), SE3.h:396-409 (float32 products, left-to-right sums)."""
    A = np.asarray(A, dtype=F).reshape(3, 4)
    B = np.asarray(B, dtype=F).reshape(3, 4)
    out = np.empty((3, 4), dtype=F)
    for i in range(3):
        for j in range(4):
            s = F(A[i, 0] * B[0, j]) + F(A[i, 1] * B[1, j])
            s = F(s + F(A[i, 2] * B[2, j]))
            if j == 3:
                s = F(s + A[i, 3])
            out[i, j] = s
    return out.reshape(12)


def invert(A):
    """SE3Invert, SE3.h:447-451."""
    A = np.asarray(A, dtype=F).reshape(3, 4)
    R = A[:, :3]
    t = A[:, 3]
    out = np.empty((3, 4), dtype=F)
    out[:, :3] = R.T
    for i in range(3):
        out[i, 3] = -F(F(F(R[0, i] * t[0]) + F(R[1, i] * t[1])) + F(R[2, i] * t[2]))
    return out.reshape(12)


def apply(T, p):
    T = np.asarray(T, dtype=F).reshape(3, 4)
    p = np.asarray(p, dtype=F)
    return (T[:, :3] @ p[..., :3].T).T + T[:, 3]


def rotate(T, v):
    T = np.asarray(T, dtype=F).reshape(3, 4)
    return (T[:, :3] @ np.asarray(v, dtype=F)[..., :3].T).T


def translation(x, y, z):
    T = identity()
    T[3], T[7], T[11] = x, y, z
    return T


def exp(t6):
    """SE3Fromse3 (REAL_SE3 branch), SE3.h:147-180."""
    p = np.asarray(t6, dtype=F)
    wx, wy, wz = p[3], p[4], p[5]
    theta = F(np.sqrt(F(wx * wx + wy * wy + wz * wz)))
    if theta == 0:
        return np.array([1, 0, 0, p[0], 0, 1, 0, p[1], 0, 0, 1, p[2]], dtype=F)
    c, s = F(np.cos(theta)), F(np.sin(theta))
    Va = F((1 - c) / (theta * theta))
    Vb = F((theta - s) / (theta * theta * theta))
    tx = (1 + Vb * (-wz * wz - wy * wy)) * p[0] + (Va * -wz + Vb * (wx * wy)) * p[1] + (Va * wy + Vb * (wx * wz)) * p[2]
    ty = (Va * wz + Vb * (wx * wy)) * p[0] + (1 + Vb * (-wz * wz - wx * wx)) * p[1] + (Va * -wx + Vb * (wy * wz)) * p[2]
    tz = (Va * -wy + Vb * (wx * wz)) * p[0] + (Va * wx + Vb * (wy * wz)) * p[1] + (1 + Vb * (-wy * wy - wx * wx)) * p[2]
    a = F(s / theta)
    b = F((1 - c) / (theta * theta))
    return np.array([
        1 + b * (-wz * wz - wy * wy), a * -wz + b * (wx * wy), a * wy + b * (wx * wz), tx,
        a * wz + b * (wx * wy), 1 + b * (-wz * wz - wx * wx), a * -wx + b * (wy * wz), ty,
        a * -wy + b * (wx * wz), a * wx + b * (wy * wz), 1 + b * (-wy * wy - wx * wx), tz], dtype=F)


def log(A):
    """se3FromSE3 (REAL_SE3 branch), SE3.h:250-299."""
    A = np.asarray(A, dtype=F)
    cos_t = F((A[0] + A[5] + A[10] - 1) / F(2))
    theta = F(0) if cos_t >= 0.9999 else F(np.arccos(cos_t))
    out = np.zeros(6, dtype=F)
    if theta == 0:
        out[0], out[1], out[2] = A[3], A[7], A[11]
        return out
    s, c = F(np.sin(theta)), F(np.cos(theta))
    wx = (A[9] - A[6]) * theta / (2 * s)
    wy = (A[2] - A[8]) * theta / (2 * s)
    wz = (A[4] - A[1]) * theta / (2 * s)
    Va = F(-0.5)
    Vb = (1 / (theta * theta)) * (1 - (s / theta) / (2 * (1 - c) / (theta * theta)))
    out[3], out[4], out[5] = wx, wy, wz
    out[0] = (1 + Vb * (-wz * wz - wy * wy)) * A[3] + (Va * -wz + Vb * (wx * wy)) * A[7] + (Va * wy + Vb * (wx * wz)) * A[11]
    out[1] = (Va * wz + Vb * (wx * wy)) * A[3] + (1 + Vb * (-wz * wz - wx * wx)) * A[7] + (Va * -wx + Vb * (wy * wz)) * A[11]
    out[2] = (Va * -wy + Vb * (wx * wz)) * A[3] + (Va * wx + Vb * (wy * wz)) * A[7] + (1 + Vb * (-wy * wy - wx * wx)) * A[11]
    return out


def from_euler(phi, theta, psi):
    """SE3FromEuler (Tait-Bryan ZYX), SE3.h:134-144."""
    cphi, sphi = np.cos(F(phi)), np.sin(F(phi))
    cth, sth = np.cos(F(theta)), np.sin(F(theta))
    cpsi, spsi = np.cos(F(psi)), np.sin(F(psi))
    return np.array([
        cth * cphi, spsi * sth * cphi - cpsi * sphi, cpsi * sth * cphi + spsi * sphi, 0,
        cth * sphi, spsi * sth * sphi + cpsi * cphi, cpsi * sth * sphi - spsi * cphi, 0,
        -sth, spsi * cth, cpsi * cth, 0], dtype=F)


def rot_x(t):
    c, s = np.cos(F(t)), np.sin(F(t))
    return np.array([1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0], dtype=F)


def rot_y(t):
    c, s = np.cos(F(t)), np.sin(F(t))
    return np.array([c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0], dtype=F)


def rot_z(t):
    c, s = np.cos(F(t)), np.sin(F(t))
    return np.array([c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0], dtype=F)


def from_dh_graspit(theta, d, a, alpha, y_rot):
    """SE3FromDHGraspit, SE3.h:103-106."""
    T = mul(rot_y(y_rot), rot_z(theta))
    T = mul(T, translation(0, 0, d))
    T = mul(T, translation(a, 0, 0))
    return mul(T, rot_x(alpha))
