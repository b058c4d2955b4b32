// dartk_fk.h -- on-device forward kinematics and the SE3 exponential (shared by dartk_model.hip and the
// device-resident Gauss-Newton loops).
#pragma once

#include "dartk_internal.h"

namespace dartk {

#ifdef __HIPCC__

// ---- forward kinematics on device -------------------------------------------------------------------
// One wave; lane 0 walks the tree (nFrames <= 65, a few hundred flops): the point is to keep the pose on
// the device so a Gauss-Newton iteration never round-trips to the host (reference: host FK + 2 blocking
// H2D copies per setPose, mirrored_model.cpp:185-225).
__device__ inline void dh_graspit(float theta, float d, float a, float alpha, float yrot, float *out) {
    // RotY(y)*RotZ(theta)*Trans(0,0,d)*Trans(a,0,0)*RotX(alpha), left-associated (SE3.h:103-106)
    const float cy = cosf(yrot), sy = sinf(yrot), ct = cosf(theta), st = sinf(theta), ca = cosf(alpha), sa = sinf(alpha);
    float RY[12] = {cy, 0, sy, 0, 0, 1, 0, 0, -sy, 0, cy, 0};
    float RZ[12] = {ct, -st, 0, 0, st, ct, 0, 0, 0, 0, 1, 0};
    float TD[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, d};
    float TA[12] = {1, 0, 0, a, 0, 1, 0, 0, 0, 0, 1, 0};
    float RX[12] = {1, 0, 0, 0, 0, ca, -sa, 0, 0, sa, ca, 0};
    float t[12];
    se3_mul(RY, RZ, t);
    se3_mul(t, TD, t);
    se3_mul(t, TA, t);
    se3_mul(t, RX, out);
}

__device__ inline void se3_exp(const float *p, float *out) {   // SE3Fromse3, SE3.h:147-180
    const float wx = p[3], wy = p[4], wz = p[5];
    const float theta = sqrtf(wx * wx + wy * wy + wz * wz);
    if (theta == 0) {
        out[0] = 1; out[1] = 0; out[2] = 0; out[3] = p[0];
        out[4] = 0; out[5] = 1; out[6] = 0; out[7] = p[1];
        out[8] = 0; out[9] = 0; out[10] = 1; out[11] = p[2];
        return;
    }
    const float Va = (1 - cosf(theta)) / (theta * theta);
    const float Vb = (theta - sinf(theta)) / (theta * theta * theta);
    const float tx = (1 + Vb * (-wz * wz - wy * wy)) * p[0] + (Va * -wz + Vb * (wx * wy)) * p[1] + (Va * wy + Vb * (wx * wz)) * p[2];
    const float ty = (Va * wz + Vb * (wx * wy)) * p[0] + (1 + Vb * (-wz * wz - wx * wx)) * p[1] + (Va * -wx + Vb * (wy * wz)) * p[2];
    const float tz = (Va * -wy + Vb * (wx * wz)) * p[0] + (Va * wx + Vb * (wy * wz)) * p[1] + (1 + Vb * (-wy * wy - wx * wx)) * p[2];
    const float a = sinf(theta) / theta;
    const float b = (1 - cosf(theta)) / (theta * theta);
    out[0] = 1 + b * (-wz * wz - wy * wy); out[1] = a * -wz + b * (wx * wy);      out[2] = a * wy + b * (wx * wz);        out[3] = tx;
    out[4] = a * wz + b * (wx * wy);       out[5] = 1 + b * (-wz * wz - wx * wx); out[6] = a * -wx + b * (wy * wz);       out[7] = ty;
    out[8] = a * -wy + b * (wx * wz);      out[9] = a * wx + b * (wy * wz);       out[10] = 1 + b * (-wy * wy - wx * wx); out[11] = tz;
}

// The constant tables one lane needs for its joint / frame: loaded ahead of time (fk_prefetch) so that the loads
// fly while the caller is still solving; nothing in them depends on the pose.
struct FkTables {
    float lo, hi, ax, ay, az, dh[4], P[12];
    int type, parent;
};

__device__ inline FkTables fk_prefetch(const ModelView &m) {
    const int lane = threadIdx.x & 63;
    FkTables t;
    if (lane + 1 < m.nFrames) {
        t.lo = m.limits[2 * lane]; t.hi = m.limits[2 * lane + 1];
        t.ax = m.jointAxis[3 * lane]; t.ay = m.jointAxis[3 * lane + 1]; t.az = m.jointAxis[3 * lane + 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) t.dh[i] = m.dh[4 * lane + i];
#pragma unroll
        for (int i = 0; i < 12; ++i) t.P[i] = m.T_pf[12 * lane + i];
        t.type = m.jointType[lane];
        t.parent = m.frameParent[lane + 1];
    } else {
        t.lo = t.hi = t.ax = t.ay = t.az = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t.dh[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 12; ++i) t.P[i] = 0.0f;
        t.type = 0; t.parent = 0;
    }
    return t;
}

// Local transform of a joint (frame joint+1) relative to its parent frame: T_pf * D(p), p = the joint value.
// mode 0: setPose (GraspIt DH form, mirrored_model.cpp:196-213), 1: setArticulation (axis-angle, :165-166)
__device__ inline void fk_local(const FkTables &t, int mode, float value, float *T) {
    const float p = fminf(fmaxf(t.lo, value), t.hi);
    float D[12];
    if (mode == 0) {
        if (t.type == 0) dh_graspit(t.dh[3] + p * t.az, t.dh[2], t.dh[0], t.dh[1] + p * t.ax, p * t.ay, D);
        else             dh_graspit(t.dh[3], t.dh[2] + p * t.az, t.dh[0] + p * t.ax, t.dh[1], p * t.ay, D);
    } else {
        float t6[6] = {0, 0, 0, 0, 0, 0};
        if (t.type == 0) { t6[3] = p * t.ax; t6[4] = p * t.ay; t6[5] = p * t.az; }
        else             { t6[0] = p * t.ax; t6[1] = p * t.ay; t6[2] = p * t.az; }
        se3_exp(t6, D);
    }
    se3_mul(t.P, D, T);
}

// Forward kinematics by ONE wave (call with all 64 lanes of a wave; nFrames <= 65). Lane j owns joint j /
// frame j+1: the trigonometry of all joints runs in parallel, then the chain products T_mf[f] = T_mf[parent]*T
// are resolved level by level of the kinematic tree (depth <= 5 for the hands). Per frame the arithmetic is the
// same sequence as the serial walk of the reference (mirrored_model.cpp:185-225), so results are identical.
// `sT` = LDS scratch of 12*nFrames floats; `value` = this lane's joint value (lane j: joint j).
__device__ inline void fk_wave(const ModelView &m, int mode, float *sT, const FkTables &t, float value) {
    const int lane = threadIdx.x & 63;
    const int f = lane + 1;
    const bool mine = f < m.nFrames;
    float T[12];
    // depth of every frame: the parent table goes through LDS (borrowing sT), the walk up the tree reads LDS
    int *sParent = reinterpret_cast<int *>(sT);
    if (lane == 0) sParent[0] = 0;
    if (mine) sParent[f] = t.parent;
    if (mine) fk_local(t, mode, value, T);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int depth = 0;
    if (mine)
        for (int q = f; q != 0; q = sParent[q]) ++depth;
    int maxDepth = depth;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxDepth = max(maxDepth, __shfl_xor(maxDepth, o));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every lane is done with sParent before sT is reused
    if (lane == 0) {
        const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < 12; ++i) { sT[i] = I[i]; m.T_mf[i] = I[i]; m.T_fm[i] = I[i]; }
    }
    for (int level = 1; level <= maxDepth; ++level) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous level's LDS writes have landed
        if (mine && depth == level) {
            float P[12], R[12], Rinv[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) P[i] = sT[12 * t.parent + i];
            se3_mul(P, T, R);
            se3_invert(R, Rinv);
#pragma unroll
            for (int i = 0; i < 12; ++i) { sT[12 * f + i] = R[i]; m.T_mf[12 * f + i] = R[i]; m.T_fm[12 * f + i] = Rinv[i]; }
        }
    }
}

// the same with the tables and the joint values read here (m.joints must be visible to the calling lane)
__device__ inline void fk_wave(const ModelView &m, int mode, float *sT) {
    const int lane = threadIdx.x & 63;
    const FkTables t = fk_prefetch(m);
    const float value = (lane + 1 < m.nFrames) ? m.joints[lane] : 0.0f;
    fk_wave(m, mode, sT, t, value);
}

#endif  // __HIPCC__

}  // namespace dartk
