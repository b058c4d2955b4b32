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

// mode 0: setPose (GraspIt DH), 1: setArticulation (axis-angle)
__device__ inline void fk_device(const ModelView &m, int mode) {
    float *T_mf = m.T_mf, *T_fm = m.T_fm;
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; ++i) { T_mf[i] = I[i]; T_fm[i] = I[i]; }
    for (int f = 1; f < m.nFrames; ++f) {
        const int joint = f - 1;
        const float p = fminf(fmaxf(m.limits[2 * joint], m.joints[joint]), m.limits[2 * joint + 1]);
        const float ax = m.jointAxis[3 * joint], ay = m.jointAxis[3 * joint + 1], az = m.jointAxis[3 * joint + 2];
        float D[12], T[12], P[12];
        if (mode == 0) {
            const float *dh = m.dh + 4 * joint;
            if (m.jointType[joint] == 0) dh_graspit(dh[3] + p * az, dh[2], dh[0], dh[1] + p * ax, p * ay, D);
            else                         dh_graspit(dh[3], dh[2] + p * az, dh[0] + p * ax, dh[1], p * ay, D);
        } else {
            float t6[6] = {0, 0, 0, 0, 0, 0};
            if (m.jointType[joint] == 0) { t6[3] = p * ax; t6[4] = p * ay; t6[5] = p * az; }
            else                         { t6[0] = p * ax; t6[1] = p * ay; t6[2] = p * az; }
            se3_exp(t6, D);
        }
        for (int i = 0; i < 12; ++i) P[i] = m.T_pf[12 * joint + i];
        se3_mul(P, D, T);
        float parentT[12];
        for (int i = 0; i < 12; ++i) parentT[i] = T_mf[12 * m.frameParent[f] + i];
        float R[12], Rinv[12];
        se3_mul(parentT, T, R);
        se3_invert(R, Rinv);
        for (int i = 0; i < 12; ++i) { T_mf[12 * f + i] = R[i]; T_fm[12 * f + i] = Rinv[i]; }
    }
}


#endif  // __HIPCC__

}  // namespace dartk
