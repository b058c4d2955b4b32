/*
 * dart_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See dart_oracle.h.
 *
 * Restates, function by function, the arithmetic of the reference's hot path. Citations are
 * file:line relative to /root/reference. Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "dart_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * small vector helpers (CUDA samples' helper_math.h semantics, which the reference includes)
 * ---------------------------------------------------------------------------------------------- */
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* SE3Transform(A, float4 b), SE3.h:411-416 */
static inline void se3_apply4(const float *A, const float *b, float *o) {
    float x = A[0] * b[0] + A[1] * b[1] + A[2] * b[2] + A[3] * b[3];
    float y = A[4] * b[0] + A[5] * b[1] + A[6] * b[2] + A[7] * b[3];
    float z = A[8] * b[0] + A[9] * b[1] + A[10] * b[2] + A[11] * b[3];
    o[0] = x; o[1] = y; o[2] = z; o[3] = b[3];
}
/* SE3Transform(A, float3 b), SE3.h:435-439 */
static inline void se3_apply3(const float *A, const float *b, float *o) {
    float x = A[0] * b[0] + A[1] * b[1] + A[2] * b[2] + A[3];
    float y = A[4] * b[0] + A[5] * b[1] + A[6] * b[2] + A[7];
    float z = A[8] * b[0] + A[9] * b[1] + A[10] * b[2] + A[11];
    o[0] = x; o[1] = y; o[2] = z;
}
/* SE3Rotate(A, float3 b), SE3.h:441-445 */
static inline void se3_rotate3(const float *A, const float *b, float *o) {
    float x = A[0] * b[0] + A[1] * b[1] + A[2] * b[2];
    float y = A[4] * b[0] + A[5] * b[1] + A[6] * b[2];
    float z = A[8] * b[0] + A[9] * b[1] + A[10] * b[2];
    o[0] = x; o[1] = y; o[2] = z;
}

void orc_se3_mul(const float *A, const float *B, float *out) {
    float r[12];
    for (int i = 0; i < 3; ++i) {
        const float *a = A + 4 * i;
        r[4 * i + 0] = a[0] * B[0] + a[1] * B[4] + a[2] * B[8];
        r[4 * i + 1] = a[0] * B[1] + a[1] * B[5] + a[2] * B[9];
        r[4 * i + 2] = a[0] * B[2] + a[1] * B[6] + a[2] * B[10];
        r[4 * i + 3] = a[0] * B[3] + a[1] * B[7] + a[2] * B[11] + a[3];
    }
    memcpy(out, r, sizeof(r));
}

void orc_se3_invert(const float *A, float *out) {
    float r[12];
    r[0] = A[0]; r[1] = A[4]; r[2] = A[8];  r[3]  = -(A[0] * A[3] + A[4] * A[7] + A[8] * A[11]);
    r[4] = A[1]; r[5] = A[5]; r[6] = A[9];  r[7]  = -(A[1] * A[3] + A[5] * A[7] + A[9] * A[11]);
    r[8] = A[2]; r[9] = A[6]; r[10] = A[10]; r[11] = -(A[2] * A[3] + A[6] * A[7] + A[10] * A[11]);
    memcpy(out, r, sizeof(r));
}

static void se3_identity(float *o) {
    static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(o, I, sizeof(I));
}

void orc_se3_exp(const float *p, float *out) {
    const float wx = p[3], wy = p[4], wz = p[5];
    const float theta = sqrtf(wx * wx + wy * wy + wz * wz);
    if (theta == 0) {
        se3_identity(out);
        out[3] = p[0]; out[7] = p[1]; out[11] = p[2];
        return;
    }
    const float Va = (1 - cosf(theta)) / (theta * theta);
    const float Vb = (theta - sinf(theta)) / (theta * theta * theta);
    const float tx = (1 + Vb * (-wz * wz - wy * wy)) * p[0] + (Va * -wz + Vb * (wx * wy)) * p[1] + (Va * wy + Vb * (wx * wz)) * p[2];
    const float ty = (Va * wz + Vb * (wx * wy)) * p[0] + (1 + Vb * (-wz * wz - wx * wx)) * p[1] + (Va * -wx + Vb * (wy * wz)) * p[2];
    const float tz = (Va * -wy + Vb * (wx * wz)) * p[0] + (Va * wx + Vb * (wy * wz)) * p[1] + (1 + Vb * (-wy * wy - wx * wx)) * p[2];
    const float a = sinf(theta) / theta;
    const float b = (1 - cosf(theta)) / (theta * theta);
    out[0] = 1 + b * (-wz * wz - wy * wy); out[1] = a * -wz + b * (wx * wy);        out[2] = a * wy + b * (wx * wz);         out[3] = tx;
    out[4] = a * wz + b * (wx * wy);       out[5] = 1 + b * (-wz * wz - wx * wx);   out[6] = a * -wx + b * (wy * wz);        out[7] = ty;
    out[8] = a * -wy + b * (wx * wz);      out[9] = a * wx + b * (wy * wz);         out[10] = 1 + b * (-wy * wy - wx * wx);  out[11] = tz;
}

void orc_se3_log(const float *A, float *a) {
    const float cosTheta = (A[0] + A[5] + A[10] - 1) / 2.0f;
    const float theta = cosTheta >= 0.9999 ? 0 : acosf(cosTheta);
    if (theta == 0) { a[3] = a[4] = a[5] = 0; a[0] = A[3]; a[1] = A[7]; a[2] = A[11]; return; }
    const float wx = (A[9] - A[6]) * theta / (2 * sinf(theta));
    const float wy = (A[2] - A[8]) * theta / (2 * sinf(theta));
    const float wz = (A[4] - A[1]) * theta / (2 * sinf(theta));
    const float Va = -0.5f;
    const float Vb = (1 / (theta * theta)) * (1 - (sinf(theta) / theta) / (2 * (1 - cosf(theta)) / (theta * theta)));
    a[3] = wx; a[4] = wy; a[5] = wz;
    a[0] = (1 + Vb * (-wz * wz - wy * wy)) * A[3] + (Va * -wz + Vb * (wx * wy)) * A[7] + (Va * wy + Vb * (wx * wz)) * A[11];
    a[1] = (Va * wz + Vb * (wx * wy)) * A[3] + (1 + Vb * (-wz * wz - wx * wx)) * A[7] + (Va * -wx + Vb * (wy * wz)) * A[11];
    a[2] = (Va * -wy + Vb * (wx * wz)) * A[3] + (Va * wx + Vb * (wy * wz)) * A[7] + (1 + Vb * (-wy * wy - wx * wx)) * A[11];
}

void orc_se3_from_euler(float phi, float theta, float psi, float *o) {
    const float cPhi = cosf(phi), sPhi = sinf(phi);
    const float cTheta = cosf(theta), sTheta = sinf(theta);
    const float cPsi = cosf(psi), sPsi = sinf(psi);
    o[0] = cTheta * cPhi; o[1] = sPsi * sTheta * cPhi - cPsi * sPhi; o[2] = cPsi * sTheta * cPhi + sPsi * sPhi; o[3] = 0;
    o[4] = cTheta * sPhi; o[5] = sPsi * sTheta * sPhi + cPsi * cPhi; o[6] = cPsi * sTheta * sPhi - sPsi * cPhi; o[7] = 0;
    o[8] = -sTheta;       o[9] = sPsi * cTheta;                      o[10] = cPsi * cTheta;                     o[11] = 0;
}

void orc_se3_from_dh_graspit(float theta, float d, float a, float alpha, float yrot, float *out) {
    /* RotY(y) * RotZ(theta) * Trans(0,0,d) * Trans(a,0,0) * RotX(alpha), left-associated (SE3.h:103-106, :215-231) */
    float RY[12] = {cosf(yrot), 0, sinf(yrot), 0, 0, 1, 0, 0, -sinf(yrot), 0, cosf(yrot), 0};
    float RZ[12] = {cosf(theta), -sinf(theta), 0, 0, sinf(theta), cosf(theta), 0, 0, 0, 0, 1, 0};
    float TD[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, d};
    float TA[12] = {1, 0, 0, a, 0, 1, 0, 0, 0, 0, 1, 0};
    float RX[12] = {1, 0, 0, 0, 0, cosf(alpha), -sinf(alpha), 0, 0, sinf(alpha), cosf(alpha), 0};
    float t[12];
    orc_se3_mul(RY, RZ, t);
    orc_se3_mul(t, TD, t);
    orc_se3_mul(t, TA, t);
    orc_se3_mul(t, RX, out);
}

/* ------------------------------------------------------------------------------------------------
 * Grid3D (grid_3d.h). Index arithmetic is done in double where the reference's literals are double.
 * ---------------------------------------------------------------------------------------------- */
void orc_grid_coords(const orc_grid *g, const float *pw, float *pg) {
    pg[0] = (pw[0] - g->offset[0]) / g->resolution;
    pg[1] = (pw[1] - g->offset[1]) / g->resolution;
    pg[2] = (pw[2] - g->offset[2]) / g->resolution;
}

int orc_grid_in_bounds_interp(const orc_grid *g, const float *p) {
    return (p[0] > 0.50001 && p[0] < (unsigned)g->dim[0] - 0.50001 &&
            p[1] > 0.50001 && p[1] < (unsigned)g->dim[1] - 0.50001 &&
            p[2] > 0.50001 && p[2] < (unsigned)g->dim[2] - 0.50001);
}

int orc_grid_in_bounds_gradient_interp(const orc_grid *g, const float *p) {
    return (p[0] > 2.50001 && p[0] < (unsigned)g->dim[0] - 2.50001 &&
            p[1] > 2.50001 && p[1] < (unsigned)g->dim[1] - 2.50001 &&
            p[2] > 2.50001 && p[2] < (unsigned)g->dim[2] - 2.50001);
}

static inline float grid_at(const orc_grid *g, int x, int y, int z) {
    return g->data[x + g->dim[0] * (y + g->dim[1] * z)];
}

/* trilinear-lookup counters by caller category: feed the algorithmic-bytes figures of bench.py */
/* Worker threads of the per-pixel / per-point loops (OpenMP). 1 (the default) is the plain serial restatement every parity
 * test runs; bench.py's all-cores CPU baseline raises it. Pixels are independent; partial systems are kept per thread over a
 * contiguous chunk and added in thread order, so a given thread count is reproducible and one thread is the serial code. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

static long long g_lookups[ORC_STAT_CATEGORIES];
static int g_cat = 0;
static int g_stats_on = 0;
#pragma omp threadprivate(g_cat)
void orc_stats_reset(void) { for (int i = 0; i < ORC_STAT_CATEGORIES; ++i) g_lookups[i] = 0; }
void orc_stats_enable(int on) { g_stats_on = on; }
void orc_stats_get(long long *out) { for (int i = 0; i < ORC_STAT_CATEGORIES; ++i) out[i] = g_lookups[i]; }
static inline int stat_enter(int cat) { int old = g_cat; g_cat = cat; return old; }

float orc_grid_value_interp(const orc_grid *g, const float *p) {
    if (g_stats_on) g_lookups[g_cat]++;   /* statistics runs are single-threaded */
    const int x0 = (int)(p[0] - 0.5); const float fx = (p[0] - 0.5) - x0;
    const int y0 = (int)(p[1] - 0.5); const float fy = (p[1] - 0.5) - y0;
    const int z0 = (int)(p[2] - 0.5); const float fz = (p[2] - 0.5) - z0;
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float dx00 = lerpf(grid_at(g, x0, y0, z0), grid_at(g, x1, y0, z0), fx);
    const float dx01 = lerpf(grid_at(g, x0, y0, z1), grid_at(g, x1, y0, z1), fx);
    const float dx10 = lerpf(grid_at(g, x0, y1, z0), grid_at(g, x1, y1, z0), fx);
    const float dx11 = lerpf(grid_at(g, x0, y1, z1), grid_at(g, x1, y1, z1), fx);
    const float dxy0 = lerpf(dx00, dx10, fy);
    const float dxy1 = lerpf(dx01, dx11, fy);
    return lerpf(dxy0, dxy1, fz);
}

void orc_grid_gradient_interp(const orc_grid *g, const float *p, float *grad) {
    float q[3];
    float fp[3], fm[3];
    for (int k = 0; k < 3; ++k) {
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
        q[k] = p[k] + 1.0f;
        fp[k] = orc_grid_value_interp(g, q);
        q[k] = p[k] - 1.0f;
        fm[k] = orc_grid_value_interp(g, q);
    }
    grad[0] = 0.5 * (fp[0] - fm[0]);
    grad[1] = 0.5 * (fp[1] - fm[1]);
    grad[2] = 0.5 * (fp[2] - fm[2]);
}

static inline orc_grid model_sdf(const orc_model *m, int s) {
    orc_grid g;
    g.dim[0] = m->sdf_dim[3 * s]; g.dim[1] = m->sdf_dim[3 * s + 1]; g.dim[2] = m->sdf_dim[3 * s + 2];
    g.offset[0] = m->sdf_offset[3 * s]; g.offset[1] = m->sdf_offset[3 * s + 1]; g.offset[2] = m->sdf_offset[3 * s + 2];
    g.resolution = m->sdf_resolution[s];
    g.data = m->sdf_data[s];
    return g;
}

/* ------------------------------------------------------------------------------------------------
 * forward kinematics
 * ---------------------------------------------------------------------------------------------- */
static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(lo, v), hi); }

void orc_fk_set_pose(const orc_model *m, const float *joints, float *T_mf, float *T_fm) {
    se3_identity(T_mf); se3_identity(T_fm);
    for (int f = 1; f < m->n_frames; ++f) {
        const int joint = f - 1;
        const float p = clampf(joints[joint], m->joint_limits[2 * joint], m->joint_limits[2 * joint + 1]);
        const float *ax = m->joint_axis + 3 * joint;
        const float *dh = m->dh + 4 * joint; /* x=a, y=alpha, z=d, w=theta */
        float D[12], T_pf[12];
        if (m->joint_type[joint] == 0) {
            orc_se3_from_dh_graspit(dh[3] + p * ax[2], dh[2], dh[0], dh[1] + p * ax[0], p * ax[1], D);
        } else {
            orc_se3_from_dh_graspit(dh[3], dh[2] + p * ax[2], dh[0] + p * ax[0], dh[1], p * ax[1], D);
        }
        orc_se3_mul(m->T_pf + 12 * joint, D, T_pf);
        orc_se3_mul(T_mf + 12 * m->frame_parent[f], T_pf, T_mf + 12 * f);
        orc_se3_invert(T_mf + 12 * f, T_fm + 12 * f);
    }
}

void orc_fk_set_articulation(const orc_model *m, const float *joints, float *T_mf, float *T_fm) {
    se3_identity(T_mf); se3_identity(T_fm);
    for (int f = 1; f < m->n_frames; ++f) {
        const int joint = f - 1;
        const float p = clampf(joints[joint], m->joint_limits[2 * joint], m->joint_limits[2 * joint + 1]);
        const float *ax = m->joint_axis + 3 * joint;
        float t6[6] = {0, 0, 0, 0, 0, 0}, D[12], T_pf[12];
        if (m->joint_type[joint] == 0) { t6[3] = p * ax[0]; t6[4] = p * ax[1]; t6[5] = p * ax[2]; }
        else                           { t6[0] = p * ax[0]; t6[1] = p * ax[1]; t6[2] = p * ax[2]; }
        orc_se3_exp(t6, D);
        orc_se3_mul(m->T_pf + 12 * joint, D, T_pf);
        orc_se3_mul(T_mf + 12 * m->frame_parent[f], T_pf, T_mf + 12 * f);
        orc_se3_invert(T_mf + 12 * f, T_fm + 12 * f);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Jacobians
 * ---------------------------------------------------------------------------------------------- */
void orc_error_jacobian(const orc_model *m, const float *T_fm, const float *T_mf, const float *pm,
                        int frame, const float *g, int articulation_only, float *J) {
    const int nJ = m->n_joints;
    float *Ja = J;
    if (!articulation_only) {
        const float e0[3] = {-1, 0, 0}, e1[3] = {0, -1, 0}, e2[3] = {0, 0, -1};
        J[0] = dot3(g, e0); J[1] = dot3(g, e1); J[2] = dot3(g, e2);
        const float r0[3] = {0, pm[2], -pm[1]}, r1[3] = {-pm[2], 0, pm[0]}, r2[3] = {pm[1], -pm[0], 0};
        J[3] = dot3(g, r0); J[4] = dot3(g, r1); J[5] = dot3(g, r2);
        Ja = J + 6;
    }
    const float p4[4] = {pm[0], pm[1], pm[2], pm[3]};
    for (int i = 0; i < nJ; ++i) {
        if (m->dependencies[frame * nJ + i] == 0) { Ja[i] = 0; continue; }
        const int jf = i + 1;
        if (m->joint_type[i] == 0) {
            float rel[4], dxa[3], dx[3];
            se3_apply4(T_fm + 12 * jf, p4, rel);
            cross3(m->joint_axis + 3 * i, rel, dxa);
            se3_rotate3(T_mf + 12 * jf, dxa, dx);
            Ja[i] = dot3(g, dx);
        } else {
            float axm[3];
            se3_rotate3(T_mf + 12 * jf, m->joint_axis + 3 * i, axm);
            Ja[i] = dot3(g, axm);
        }
    }
}

void orc_model_jacobian_3d(const orc_model *m, const float *T_fm, const float *T_mf, const float *pm,
                           int frame, float *J3D) {
    const int nJ = m->n_joints;
    const float b[18] = {-1, 0, 0, 0, -1, 0, 0, 0, -1, 0.0f, pm[2], -pm[1], -pm[2], 0.0f, pm[0], pm[1], -pm[0], 0.0f};
    memcpy(J3D, b, sizeof(b));
    const float p4[4] = {pm[0], pm[1], pm[2], pm[3]};
    for (int joint = 0; joint < nJ; ++joint) {
        float *o = J3D + 3 * (6 + joint);
        if (!m->dependencies[frame * nJ + joint]) { o[0] = o[1] = o[2] = 0; continue; }
        const int jf = joint + 1;
        if (m->joint_type[joint] == 0) {
            float rel[4], da[3];
            se3_apply4(T_fm + 12 * jf, p4, rel);
            cross3(m->joint_axis + 3 * joint, rel, da);
            se3_rotate3(T_mf + 12 * jf, da, o);
        } else {
            /* model.cpp:481-488 rotates the axis by T_pa (joint position/orientation) then by the parent's
             * frame-to-model transform; T_pa == T_pf for the se3-orientation (version 0) convention. */
            float axp[3];
            se3_rotate3(m->T_pf + 12 * joint, m->joint_axis + 3 * joint, axp);
            se3_rotate3(T_mf + 12 * m->frame_parent[jf], axp, o);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * obs -> mod data association
 * ---------------------------------------------------------------------------------------------- */
/* one pixel of gpu_errorAndDataAssociationObsToMod (obsToMod.cu:12-114): returns 1 when the pixel is associated */
static int assoc_single_pixel(int index, const float *vert, const float *norm, const orc_model *m, const float *T_mc,
                              const float *T_fm, float distanceThreshold, float normThreshold, float planeOffset,
                              const float *planeNormal, int *gridOut, float *errOut, int32_t *dbgDA, float *dbgErr,
                              float *dbgNorm) {
    stat_enter(ORC_STAT_ASSOC);
    if (dbgDA) dbgDA[index] = -1;
    if (dbgErr) dbgErr[index] = NAN;
    if (dbgNorm) { dbgNorm[4 * index] = dbgNorm[4 * index + 1] = dbgNorm[4 * index + 2] = dbgNorm[4 * index + 3] = 0; }
    const float *xc = vert + 4 * index;
    if (!(xc[3] > 0)) return 0;
    float xm[4];
    se3_apply4(T_mc, xc, xm);
    if (!(dot3(xm, planeNormal) >= planeOffset)) return 0;

    float sdfError = 1e20f;
    int grid = -1;
    for (int g = 0; g < m->n_sdfs; ++g) {
        const int f = m->sdf_frame[g];
        float xf[4], xg[3];
        se3_apply4(T_fm + 12 * f, xm, xf);
        const orc_grid sdf = model_sdf(m, g);
        orc_grid_coords(&sdf, xf, xg);
        if (!orc_grid_in_bounds_gradient_interp(&sdf, xg)) continue;
        const float d = orc_grid_value_interp(&sdf, xg) * sdf.resolution;
        if (d < sdfError) { sdfError = d; grid = g; }
    }
    if (sdfError * sdfError > distanceThreshold * distanceThreshold) return 0;

    {
        const int f = m->sdf_frame[grid];
        float xf[4], xg[3], gr[3];
        se3_apply4(T_fm + 12 * f, xm, xf);
        const orc_grid sdf = model_sdf(m, grid);
        orc_grid_coords(&sdf, xf, xg);
        stat_enter(ORC_STAT_ASSOC_GRAD);
        orc_grid_gradient_interp(&sdf, xg, gr);
        stat_enter(ORC_STAT_ASSOC);
        /* nPred = -1 * (SE3Invert(T_fm[f]*T_mc) * normalize(float4(grad,0))), obsToMod.cu:84 */
        float T_fc[12], T_cf[12];
        orc_se3_mul(T_fm + 12 * f, T_mc, T_fc);
        orc_se3_invert(T_fc, T_cf);
        float g4[4] = {gr[0], gr[1], gr[2], 0};
        const float invLen = 1.0f / sqrtf(g4[0] * g4[0] + g4[1] * g4[1] + g4[2] * g4[2] + g4[3] * g4[3]);
        g4[0] *= invLen; g4[1] *= invLen; g4[2] *= invLen; g4[3] *= invLen;
        float nPred[4];
        se3_apply4(T_cf, g4, nPred);
        nPred[0] = -1 * nPred[0]; nPred[1] = -1 * nPred[1]; nPred[2] = -1 * nPred[2]; nPred[3] = -1 * nPred[3];
        if (dbgNorm) memcpy(dbgNorm + 4 * index, nPred, 4 * sizeof(float));
        const float *v = norm + 4 * index;
        if (v[3] > 0.0) {
            const float d = nPred[0] * v[0] + nPred[1] * v[1] + nPred[2] * v[2] + nPred[3] * 0.0f;
            if (d < normThreshold) return 0;
        }
    }
    if (dbgDA) dbgDA[index] = grid;
    if (dbgErr) dbgErr[index] = sdfError;
    *gridOut = grid; *errOut = sdfError;
    return 1;
}

int orc_assoc_single(const float *vert, const float *norm, int width, int height, const orc_model *m,
                     const float *T_mc, const float *T_fm, float distanceThreshold, float normThreshold,
                     float planeOffset, const float *planeNormal, orc_da_point *pts,
                     int32_t *dbgDA, float *dbgErr, float *dbgNorm) {
    int count = 0;
    const int oldcat = stat_enter(ORC_STAT_ASSOC);
    const int nPix = width * height;
    if (g_threads <= 1) {
        for (int index = 0; index < nPix; ++index) {
            int grid; float err;
            if (!assoc_single_pixel(index, vert, norm, m, T_mc, T_fm, distanceThreshold, normThreshold, planeOffset, planeNormal,
                                    &grid, &err, dbgDA, dbgErr, dbgNorm)) continue;
            pts[count].index = index; pts[count].dataAssociation = grid; pts[count].error = err;
            ++count;
        }
    } else {    /* pixels are independent; the list keeps pixel order (the reference's atomic append has no order) */
        int *grids = (int *)malloc(sizeof(int) * (size_t)nPix);
        float *errs = (float *)malloc(sizeof(float) * (size_t)nPix);
#pragma omp parallel for schedule(static, 1024) num_threads(g_threads)
        for (int index = 0; index < nPix; ++index) {
            grids[index] = -1;
            int grid; float err;
            if (assoc_single_pixel(index, vert, norm, m, T_mc, T_fm, distanceThreshold, normThreshold, planeOffset, planeNormal,
                                   &grid, &err, dbgDA, dbgErr, dbgNorm)) { grids[index] = grid; errs[index] = err; }
        }
        for (int index = 0; index < nPix; ++index) {
            if (grids[index] < 0) continue;
            pts[count].index = index; pts[count].dataAssociation = grids[index]; pts[count].error = errs[index];
            ++count;
        }
        free(grids); free(errs);
    }
    stat_enter(oldcat);
    return count;
}

/* one pixel of the multi-model association (obsToMod.cu:139-207): returns the model (-1 none), *gridOut, *errOut */
static int assoc_multi_pixel(int index, const float *vert, int nModels, const orc_model *const *models, const float *T_mcs,
                             const float *const *T_fms, const float *distanceThresholds, const float *planeOffsets,
                             const float *planeNormals, int *gridOut, float *errOut) {
    const float *xc = vert + 4 * index;
    if (!(xc[3] > 0)) return -1;
    float sdfError = 1e20f;
    int aModel = -1, aGrid = -1;
    for (int m = 0; m < nModels; ++m) {
        float xm[4];
        se3_apply4(T_mcs + 12 * m, xc, xm);
        if (!(dot3(xm, planeNormals + 3 * m) >= planeOffsets[m])) continue;
        const orc_model *mod = models[m];
        for (int g = 0; g < mod->n_sdfs; ++g) {
            const int f = mod->sdf_frame[g];
            float xf[4], xg[3];
            se3_apply4(T_fms[m] + 12 * f, xm, xf);
            const orc_grid sdf = model_sdf(mod, g);
            orc_grid_coords(&sdf, xf, xg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, xg)) continue;
            const float d = orc_grid_value_interp(&sdf, xg) * sdf.resolution;
            if (d < sdfError) {
                if (d * d < distanceThresholds[m] * distanceThresholds[m]) {
                    sdfError = d; aGrid = g; aModel = m;
                }
            }
        }
    }
    *gridOut = aGrid; *errOut = sdfError;
    return aModel;
}

void orc_assoc_multi(const float *vert, const float *norm, int width, int height, int nModels,
                     const orc_model *const *models, const float *T_mcs, const float *const *T_fms,
                     const float *distanceThresholds, const float *planeOffsets, const float *planeNormals,
                     orc_da_point *const *pts, int32_t *counts, int32_t *dbgDA, float *dbgErr) {
    (void)norm;
    const int nPix = width * height;
    for (int m = 0; m < nModels; ++m) counts[m] = 0;
    const int oldcat = stat_enter(ORC_STAT_ASSOC);
    int *amodel = NULL, *agrid = NULL;
    float *aerr = NULL;
    if (g_threads > 1) {        /* pixels are independent: decisions in parallel, the lists appended in pixel order afterwards */
        amodel = (int *)malloc(sizeof(int) * (size_t)nPix);
        agrid = (int *)malloc(sizeof(int) * (size_t)nPix);
        aerr = (float *)malloc(sizeof(float) * (size_t)nPix);
#pragma omp parallel for schedule(static, 1024) num_threads(g_threads)
        for (int index = 0; index < nPix; ++index)
            amodel[index] = assoc_multi_pixel(index, vert, nModels, models, T_mcs, T_fms, distanceThresholds, planeOffsets, planeNormals,
                                              &agrid[index], &aerr[index]);
    }
    for (int index = 0; index < nPix; ++index) {
        if (dbgDA) dbgDA[index] = -1;
        if (dbgErr) dbgErr[index] = NAN;
        int aGrid = -1; float sdfError = 1e20f;
        int aModel;
        if (amodel) { aModel = amodel[index]; aGrid = agrid[index]; sdfError = aerr[index]; }
        else aModel = assoc_multi_pixel(index, vert, nModels, models, T_mcs, T_fms, distanceThresholds, planeOffsets, planeNormals, &aGrid, &sdfError);
        if (aModel != -1) {
            if (dbgDA) dbgDA[index] = ((aModel << 16) | aGrid);
            if (dbgErr) dbgErr[index] = sdfError;
            orc_da_point *p = pts[aModel] + counts[aModel]++;
            p->index = index; p->dataAssociation = aGrid; p->error = sdfError;
        }
    }
    free(amodel); free(agrid); free(aerr);
    stat_enter(oldcat);
}

/* ------------------------------------------------------------------------------------------------
 * normal equations: packing (kernel_common.h:141-153) with double accumulators
 * ---------------------------------------------------------------------------------------------- */
static inline int jtj_size(int d) { return (d * (d + 1)) >> 1; }

typedef struct { int dims; double *acc; } sys_acc;

static sys_acc sys_new(int dims) {
    sys_acc s; s.dims = dims;
    s.acc = (double *)calloc((size_t)(dims + jtj_size(dims) + 1), sizeof(double));
    return s;
}
static void sys_emit(sys_acc *s, float *result) {
    const int n = s->dims + jtj_size(s->dims) + 1;
    for (int i = 0; i < n; ++i) result[i] = (float)s->acc[i];
    free(s->acc);
}
/* computeSquaredLossResult, kernel_common.h:141-153 */
static void sys_add_squared(sys_acc *s, float residual, const float *J) {
    const int dims = s->dims;
    double *JTr = s->acc, *JTJ = s->acc + dims, *e = s->acc + dims + jtj_size(dims);
    for (int i = 0; i < dims; ++i) {
        if (J[i] == 0.0f) continue;
        float v = J[i] * residual;
        JTr[i] += v;
        for (int j = 0; j <= i; ++j) {
            float v2 = J[i] * J[j];
            JTJ[((i * (i + 1)) >> 1) + j] += v2;
        }
    }
    *e += (float)(0.5 * residual * residual);
}

/* doPoseGradientReduction / doParamMapping (+ArticulationOnly), kernel_common.h:73-139.
 * off = 6 for full-pose systems, 0 for articulation-only. */
static void reduce_jacobian(int reduction, int off, int fullDims, int redDims, const float *de,
                            const float *A, const int32_t *mapping, float *J) {
    const int nf = fullDims - 6, nr = redDims - 6;
    for (int r = 0; r < off; ++r) J[r] = de[r];
    if (reduction == 1) {
        for (int r = 0; r < nr; ++r) {
            J[off + r] = 0;
            for (int f = 0; f < nf; ++f) {
                float mult = A[r + f * nr];
                if (mult == 0.f) continue;
                J[off + r] += de[off + f] * mult;
            }
        }
    } else {
        for (int r = 0; r < nr; ++r) J[off + r] = 0;
        for (int f = 0; f < nf; ++f) J[off + mapping[f]] += de[off + f];
    }
}

void orc_ne_obs2mod(const float *vert, const orc_model *m, const float *T_mc, const float *T_fm,
                    const float *T_mf, const orc_da_point *pts, int nPoints, float huberDelta,
                    int reduction, int redDims, const float *dtheta_dalpha, const int32_t *mapping,
                    float *result) {
    const int fullDims = 6 + m->n_joints;
    const int dims = reduction ? redDims : fullDims;
    sys_acc s = sys_new(dims);
    const int oldcat = stat_enter(ORC_STAT_NE_OBS2MOD);
    if (nPoints > 10) { /* obsToMod.cu:646,723,783 */
        const int nT = g_threads > 1 ? g_threads : 1;
        const int nSys = dims + jtj_size(dims) + 1;
        double *part = nT > 1 ? (double *)calloc((size_t)nT * nSys, sizeof(double)) : NULL;
#pragma omp parallel num_threads(nT) if (nT > 1)
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            stat_enter(ORC_STAT_NE_OBS2MOD);
            sys_acc st = s;
            if (nT > 1) st.acc = part + (size_t)tid * nSys;
            const int k0 = (int)((long long)nPoints * tid / nT), k1 = (int)((long long)nPoints * (tid + 1) / nT);
            float *de = (float *)malloc(sizeof(float) * (size_t)(fullDims + dims));
            float *Jr = de + fullDims;
            for (int k = k0; k < k1; ++k) {
                float xm[4], xf[4], xg[3], gf[3], gm[3];
                se3_apply4(T_mc, vert + 4 * pts[k].index, xm);
                const int g = pts[k].dataAssociation;
                const int obsFrame = m->sdf_frame[g];
                se3_apply4(T_fm + 12 * obsFrame, xm, xf);
                const orc_grid sdf = model_sdf(m, g);
                orc_grid_coords(&sdf, xf, xg);
                orc_grid_gradient_interp(&sdf, xg, gf);
                se3_rotate3(T_mf + 12 * obsFrame, gf, gm);
                orc_error_jacobian(m, T_fm, T_mf, xm, obsFrame, gm, 0, de);
                const float *J = de;
                if (reduction) { reduce_jacobian(reduction, 6, fullDims, redDims, de, dtheta_dalpha, mapping, Jr); J = Jr; }
                const float err = pts[k].error;
                if (fabsf(err) < huberDelta) {
                    sys_add_squared(&st, -err, J);
                } else { /* Huber outlier branch, obsToMod.cu:318-332 (full) / :404-415 (reduced: no JTJ) */
                    float v = huberDelta;
                    if (err < 0) v = -v;
                    double *JTr = st.acc, *JTJ = st.acc + dims, *e = st.acc + dims + jtj_size(dims);
                    for (int i = 0; i < dims; ++i) {
                        if (J[i] == 0.0f) continue;
                        JTr[i] += (float)(v * -J[i]);
                        if (!reduction) {
                            for (int j = 0; j <= i; ++j) { float v2 = J[i] * J[j]; JTJ[((i * (i + 1)) >> 1) + j] += v2; }
                        }
                    }
                    *e += (float)(huberDelta * (fabsf(err) - 0.5 * huberDelta));
                }
            }
            free(de);
        }
        if (part) {
            for (int t = 0; t < nT; ++t) for (int i = 0; i < nSys; ++i) s.acc[i] += part[(size_t)t * nSys + i];
            free(part);
        }
    }
    stat_enter(oldcat);
    sys_emit(&s, result);
}

/* ------------------------------------------------------------------------------------------------
 * SDF ray casting (raycast.cu:11-35, 121-215)
 * ---------------------------------------------------------------------------------------------- */
static int intersect_box(const float *rayDir, const float *origin, const float *bmin, const float *bmax, float *t) {
    float invR[3], tbot[3], ttop[3], tmin[3], tmax[3];
    for (int k = 0; k < 3; ++k) {
        invR[k] = 1.0f / rayDir[k];           /* helper_math: float / float3 */
        tbot[k] = invR[k] * (bmin[k] - origin[k]);
        ttop[k] = invR[k] * (bmax[k] - origin[k]);
        tmin[k] = fminf(ttop[k], tbot[k]);
        tmax[k] = fmaxf(ttop[k], tbot[k]);
    }
    float largest_tmin = fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]);
    float smallest_tmax = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
    t[0] = largest_tmin; t[1] = smallest_tmax;
    return smallest_tmax > largest_tmin;
}

void orc_raycast_prediction(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                            const orc_model *m, const float *T_mc, const float *T_fm, const float *T_mf,
                            float levelSet, float *prediction) {
    float T_cm[12];
    orc_se3_invert(T_mc, T_cm);
    const int oldcat = stat_enter(ORC_STAT_RAYCAST);
#pragma omp parallel for schedule(dynamic, 2) num_threads(g_threads) if (g_threads > 1)
    for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
        stat_enter(ORC_STAT_RAYCAST);
        const int index = x + y * width;
        float *pr = prediction + 4 * index;
        if (modelNum == 0) { pr[2] = 0; pr[3] = -1; }
        const float invflx = 1.0f / flx, invfly = 1.0f / fly;
        const float uvx = x + 0.5, uvy = y + 0.5;
        float rd[3] = {(uvx - ppx) * invflx, (uvy - ppy) * invfly, 1};
        {
            const float inv = 1.0f / sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
            rd[0] *= inv; rd[1] *= inv; rd[2] *= inv;
        }
        float closestT = 10000;
        for (int s = 0; s < m->n_sdfs; ++s) {
            const int f = m->sdf_frame[s];
            const orc_grid sdf = model_sdf(m, s);
            float T_fc[12];
            orc_se3_mul(T_fm + 12 * f, T_mc, T_fc);
            const float o4[4] = {0, 0, 0, 1}, d4[4] = {rd[0], rd[1], rd[2], 0};
            float of[4], df[4], dn[3];
            se3_apply4(T_fc, o4, of);
            se3_apply4(T_fc, d4, df);
            {
                const float inv = 1.0f / sqrtf(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
                dn[0] = df[0] * inv; dn[1] = df[1] * inv; dn[2] = df[2] * inv;
            }
            float smin[3], smax[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                smin[k] = sdf.offset[k];
                smax[k] = sdf.offset[k] + sdf.resolution * (float)(unsigned)sdf.dim[k];
                hi[k] = (float)((unsigned)sdf.dim[k] - 2u);
            }
            float t[2];
            if (!intersect_box(dn, of, smin, smax, t)) continue;
            float xf[3], xg[3];
#define ORC_RC_EVAL(tt)                                                                          \
            for (int k = 0; k < 3; ++k) {                                                        \
                xf[k] = dn[k] * (tt) + of[k];                                                    \
                xg[k] = fmaxf(1.0f, fminf((xf[k] - sdf.offset[k]) / sdf.resolution, hi[k]));     \
            }
            ORC_RC_EVAL(t[0]);
            float sdfVal = orc_grid_value_interp(&sdf, xg) * sdf.resolution;
            float sdfValPrev = sdfVal;
            float tPrev = t[0];
            while (t[0] < t[1] && t[0] < closestT) {
                ORC_RC_EVAL(t[0]);
                sdfVal = orc_grid_value_interp(&sdf, xg) * sdf.resolution;
                if (sdfValPrev > levelSet && sdfVal <= levelSet) {
                    const float tHit = (t[0] - tPrev) * (levelSet - sdfValPrev) / (sdfVal - sdfValPrev) + tPrev;
                    float xh[4] = {dn[0] * tHit + of[0], dn[1] * tHit + of[1], dn[2] * tHit + of[2], 1};
                    float T_cf[12], xc[4];
                    orc_se3_mul(T_cm, T_mf + 12 * f, T_cf);
                    se3_apply4(T_cf, xh, xc);
                    const float currentZ = pr[2];
                    if (currentZ == 0 || xc[2] < currentZ) {
                        const int id = (modelNum << 16) + s;
                        pr[0] = xc[0]; pr[1] = xc[1]; pr[2] = xc[2]; pr[3] = (float)id;
                    }
                    closestT = tHit;
                }
                sdfValPrev = sdfVal;
                tPrev = t[0];
                t[0] += fmaxf(sdf.resolution / 10, fabsf(sdfVal));
            }
#undef ORC_RC_EVAL
        }
    }
    stat_enter(oldcat);
}

void orc_cull_unobservable(float *pred, int pw, int ph, const float *obs, int ow, int oh) {
    for (int py = 0; py < ph; ++py) for (int px = 0; px < pw; ++px) {
        const int ox = px * ow / pw, oy = py * oh / ph;
        const int oi = ox + oy * ow;
        if (obs[4 * oi + 3] <= 0 || obs[4 * (oi + 1) + 3] <= 0 || obs[4 * (oi + ow) + 3] <= 0 ||
            obs[4 * (oi + ow + 1) + 3] <= 0) {
            pred[4 * (px + py * pw) + 2] = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * sensor front end: depth -> vertices -> normals (the step before the path)
 * ---------------------------------------------------------------------------------------------- */
/* gpu_depthToVertices<float>, organized_point_cloud.cu:11-40. An out-of-range depth only clears .w (:36-38): x, y, z of
 * that pixel keep what vertOut held before the call. */
void orc_depth_to_vertices(const float *depth, float *vert, int width, int height, float ppx, float ppy, float flx,
                           float fly, float rangeMin, float rangeMax) {
    for (int v = 0; v < height; ++v) for (int u = 0; u < width; ++u) {
        const int index = u + v * width;
        const float d = depth[index];
        float *o = vert + 4 * index;
        if (d >= rangeMin && d <= rangeMax) {
            o[0] = ((float)u - ppx) * (d / flx);   /* :30 (int - float promotes the int) */
            o[1] = ((float)v - ppy) * (d / fly);   /* :31 */
            o[2] = d;
            o[3] = 1.0f;
        } else {
            o[3] = 0;
        }
    }
}

/* gpu_verticesToNormals, organized_point_cloud.cu:265-305: central differences over the 4-neighbourhood, a neighbour
 * outside the image or with w == 0 replaced by the pixel itself (:286-292), n = cross(vY, vX) normalised, w = 1;
 * invalid pixels and degenerate crosses (len2 == 0, NaN) give (0,0,0,0). */
void orc_vertices_to_normals(const float *vert, float *norm, int width, int height) {
    for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
        const int index = x + y * width;
        const float *v = vert + 4 * index;
        float *o = norm + 4 * index;
        if (v[3] == 0) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
        const float *vL = vert + 4 * (x == 0 ? index : index - 1);
        const float *vR = vert + 4 * (x == width - 1 ? index : index + 1);
        const float *vU = vert + 4 * (y == 0 ? index : index - width);
        const float *vD = vert + 4 * (y == height - 1 ? index : index + width);
        if (vR[3] == 0) vR = v;
        if (vL[3] == 0) vL = v;
        if (vD[3] == 0) vD = v;
        if (vU[3] == 0) vU = v;
        const float vX[3] = {vR[0] - vL[0], vR[1] - vL[1], vR[2] - vL[2]};
        const float vY[3] = {vD[0] - vU[0], vD[1] - vU[1], vD[2] - vU[2]};
        /* helper_math.h cross(a,b) = (a.y b.z - a.z b.y, a.z b.x - a.x b.z, a.x b.y - a.y b.x) */
        const float n[3] = {vY[1] * vX[2] - vY[2] * vX[1], vY[2] * vX[0] - vY[0] * vX[2], vY[0] * vX[1] - vY[1] * vX[0]};
        const float len2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (len2 > 0) {
            const float invLen = 1.0f / sqrtf(len2);
            o[0] = n[0] * invLen; o[1] = n[1] * invLen; o[2] = n[2] * invLen; o[3] = 1;
        } else {
            o[0] = o[1] = o[2] = o[3] = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * observation SDF: splat + distance transform
 * ---------------------------------------------------------------------------------------------- */
void orc_splat_obs_sdf(const float *vert, int width, int height, const float *T_cm, const orc_grid *g,
                       float focalLength, float *data) {
    const float truncVal = 1000.0f;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
    for (int z = 0; z < g->dim[2]; ++z) for (int y = 0; y < g->dim[1]; ++y) for (int x = 0; x < g->dim[0]; ++x) {
        float p[3] = {g->offset[0] + g->resolution * (float)x, g->offset[1] + g->resolution * (float)y,
                      g->offset[2] + g->resolution * (float)z};
        float c[3];
        se3_apply3(T_cm, p, c);
        /* round() of a float expression promoted to double; (int) of the rounded value (modToObs.cu:421-422) */
        const int u = (int)round((focalLength / c[2]) * c[0] + (width >> 1));
        const int v = (int)round((focalLength / c[2]) * c[1] + (height >> 1));
        float *sv = data + (x + g->dim[0] * (y + g->dim[1] * z));
        if (u < 0 || u >= width || v < 0 || v >= height) *sv = truncVal;
        else if (vert[4 * (u + v * width) + 3] == 0 || vert[4 * (u + v * width) + 2] == 0) *sv = 0.5 * truncVal;
        else if (vert[4 * (u + v * width) + 2] < c[2]) *sv = 0;
        else *sv = truncVal;
    }
}

#define ORC_INF 1e20 /* distance_transforms.h:4 (a double literal assigned to Real=float) */

/* one 1-D lower-envelope pass (distance_transforms.cu:54-78) */
static void dt_1d(const float *fIn, float *fOut, int len, int stride, int takeSqrt, float *z, int *v) {
    int k = 0;
    v[0] = 0;
    z[0] = -ORC_INF;
    z[1] = +ORC_INF;
    for (int q = 1; q <= len - 1; q++) {
        float s = ((fIn[q * stride] + q * q) - (fIn[v[k] * stride] + v[k] * v[k])) / (2 * q - 2 * v[k]);
        while (s <= z[k]) {
            k--;
            s = ((fIn[q * stride] + q * q) - (fIn[v[k] * stride] + v[k] * v[k])) / (2 * q - 2 * v[k]);
        }
        k++;
        v[k] = q;
        z[k] = s;
        z[k + 1] = +ORC_INF;
    }
    k = 0;
    for (int q = 0; q <= len - 1; q++) {
        while (z[k + 1] < q) k++;
        if (takeSqrt) fOut[q * stride] = sqrtf((q - v[k]) * (q - v[k]) + fIn[v[k] * stride]);
        else fOut[q * stride] = (q - v[k]) * (q - v[k]) + fIn[v[k] * stride];
    }
}

void orc_distance_transform_3d(float *in, float *out, int width, int height, int depth, int takeSqrt) {
    int maxLen = width > height ? width : height; if (depth > maxLen) maxLen = depth;
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
        float *z = (float *)malloc(sizeof(float) * (size_t)(maxLen + 1));
        int *v = (int *)malloc(sizeof(int) * (size_t)maxLen);
        /* x-direction: in -> out (rows = height*depth, outer stride width, inner 1) */
#pragma omp for
        for (int t = 0; t < height * depth; ++t) dt_1d(in + (size_t)t * width, out + (size_t)t * width, width, 1, 0, z, v);
        /* z-direction: out -> in (num = width*height, outer stride 1, inner width*height) */
#pragma omp for
        for (int t = 0; t < width * height; ++t) dt_1d(out + t, in + t, depth, width * height, 0, z, v);
        /* y-direction: in -> out (a<width, b<depth; strides 1, width*height; inner width) */
#pragma omp for
        for (int b = 0; b < depth; ++b) for (int a = 0; a < width; ++a)
            dt_1d(in + a + (size_t)b * width * height, out + a + (size_t)b * width * height, height, width, takeSqrt, z, v);
        free(z); free(v);
    }
}

void orc_signed_distance_transform_3d(float *in, float *out, int width, int height, int depth, int takeSqrt) {
    const size_t n = (size_t)width * height * depth;
    float *seed = out;
    /* gpu_seedInverseTransform3D, distance_transforms.cu:173-227 */
    for (int z = 0; z < depth; ++z) for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
        const size_t index = x + (size_t)width * (y + (size_t)height * z);
        if (in[index] == 0) {
            seed[index] = ORC_INF;
            if ((x > 0 && in[index - 1] != 0.0) || (x < width - 1 && in[index + 1] != 0.0) ||
                (y > 0 && in[index - width] != 0.0) || (y < height - 1 && in[index + width] != 0.0) ||
                (z > 0 && in[index - (size_t)width * height] != 0.0) ||
                (z < depth - 1 && in[index + (size_t)width * height] != 0.0)) {
                seed[index] = 0.0;
            }
        } else {
            seed[index] = 0;
        }
    }
    float *seedDT = (float *)malloc(sizeof(float) * n);
    orc_distance_transform_3d(seed, seedDT, width, height, depth, takeSqrt);
    orc_distance_transform_3d(in, out, width, height, depth, takeSqrt); /* clobbers 'in' with the x/z partial */
    /* gpu_combineTransforms3D (:248-263) reads mask = in AFTER it was clobbered: the partial squared
     * distance is 0 exactly where the voxel itself was foreground, so the mask survives. */
    for (size_t i = 0; i < n; ++i) if (in[i] == 0) out[i] = -seedDT[i];
    free(seedDT);
}

/* ------------------------------------------------------------------------------------------------
 * mod -> obs normal equations
 * ---------------------------------------------------------------------------------------------- */
int orc_ne_mod2obs(const float *pred, int width, int height, int modelNum, const orc_model *m,
                   const float *T_gc, const float *T_fm, const float *T_mf, const orc_grid *obsSdf,
                   float truncationDist, int reduction, int redDims, const float *dtheta_dalpha,
                   const int32_t *mapping, float *result) {
    const int fullDims = 6 + m->n_joints;
    const int dims = reduction ? redDims : fullDims;
    sys_acc s = sys_new(dims);
    int numPredictions = 0;
    const int oldcat = stat_enter(ORC_STAT_NE_MOD2OBS);
    const int nT = g_threads > 1 ? g_threads : 1;
    const int nSys = dims + jtj_size(dims) + 1;
    const int nPix = width * height;
    double *part = nT > 1 ? (double *)calloc((size_t)nT * nSys, sizeof(double)) : NULL;
#pragma omp parallel num_threads(nT) if (nT > 1) reduction(+ : numPredictions)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        stat_enter(ORC_STAT_NE_MOD2OBS);
        sys_acc st = s;
        if (nT > 1) st.acc = part + (size_t)tid * nSys;
        float *de = (float *)malloc(sizeof(float) * (size_t)(fullDims + dims));
        float *Jr = de + fullDims;
        /* interleaved blocks of rows so that the threads share the (compact) silhouette evenly; per thread in pixel order */
        for (int index = 0; index < nPix; ++index) {
            if (nT > 1 && ((index / (width > 0 ? width : 1)) % nT) != tid) continue;
            const float *pc = pred + 4 * index;
            if (pc[2] == 0) continue;
            float pm[4], pg[3];
            se3_apply3(T_gc, pc, pm); pm[3] = 1;
            orc_grid_coords(obsSdf, pm, pg);
            if (!orc_grid_in_bounds_gradient_interp(obsSdf, pg)) continue;
            const float residual = orc_grid_value_interp(obsSdf, pg) * obsSdf->resolution;
            if (truncationDist >= 0 && (residual >= truncationDist || residual < 0)) continue;
            const int label = (int)round(pc[3]);
            const int model = label >> 16;
            const int sdf = label & 65535;
            if (model != modelNum) continue;
            const int predFrame = m->sdf_frame[sdf];
            float gm[3];
            orc_grid_gradient_interp(obsSdf, pg, gm);
            orc_error_jacobian(m, T_fm, T_mf, pm, predFrame, gm, 0, de);
            const float *J = de;
            if (reduction) { reduce_jacobian(reduction, 6, fullDims, redDims, de, dtheta_dalpha, mapping, Jr); J = Jr; }
            ++numPredictions;
            sys_add_squared(&st, residual, J);
        }
        free(de);
    }
    if (part) {
        for (int t = 0; t < nT; ++t) for (int i = 0; i < nSys; ++i) s.acc[i] += part[(size_t)t * nSys + i];
        free(part);
    }
    stat_enter(oldcat);
    sys_emit(&s, result);
    return numPredictions;
}

float orc_error_mod2obs(const float *pred, int width, int height, const orc_grid *obsSdf) {
    double acc = 0;
    for (int index = 0; index < width * height; ++index) {
        const float *pv = pred + 4 * index;
        if (pv[2] == 0) continue;
        float pg[3];
        orc_grid_coords(obsSdf, pv, pg);
        if (!orc_grid_in_bounds_gradient_interp(obsSdf, pg)) continue;
        const float err = orc_grid_value_interp(obsSdf, pg) * obsSdf->resolution;
        acc += (float)(0.5 * err * err);
    }
    return (float)acc;
}

/* ------------------------------------------------------------------------------------------------
 * intersection terms
 * ---------------------------------------------------------------------------------------------- */
int orc_ne_self_intersection(const float *sites, int nSites, const orc_model *m, const float *T_fm,
                             const float *T_mf, const int32_t *pot, int reduction, int redDims,
                             const float *dtheta_dalpha, const int32_t *mapping, float *result) {
    const int oldcat_ = stat_enter(ORC_STAT_INTERSECTION);
    const int fullDims = 6 + m->n_joints;
    const int adims = (reduction ? redDims : fullDims) - 6;
    sys_acc s = sys_new(adims);
    float *de = (float *)malloc(sizeof(float) * (size_t)(fullDims + adims + 1));
    float *Jr = de + fullDims;
    int nColl = 0;
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        const int srcFrame = m->sdf_frame[srcGrid];
        float vm[4];
        se3_apply4(T_mf + 12 * srcFrame, vs, vm);
        for (int dstGrid = 0; dstGrid < m->n_sdfs; ++dstGrid) {
            if (!pot[dstGrid + srcGrid * m->n_sdfs]) continue;
            const int dstFrame = m->sdf_frame[dstGrid];
            float vd[4], vg[3];
            se3_apply4(T_fm + 12 * dstFrame, vm, vd);
            const orc_grid sdf = model_sdf(m, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            const float residual = orc_grid_value_interp(&sdf, vg) * sdf.resolution;
            if (residual < 0) {
                float gf[3], gm[3];
                orc_grid_gradient_interp(&sdf, vg, gf);
                se3_rotate3(T_mf + 12 * dstFrame, gf, gm);
                orc_error_jacobian(m, T_fm, T_mf, vm, srcFrame, gm, 1, de);
                const float *J = de;
                if (reduction) { reduce_jacobian(reduction, 0, fullDims, redDims, de, dtheta_dalpha, mapping, Jr); J = Jr; }
                sys_add_squared(&s, residual, J);
                ++nColl;
                break; /* 'return' after the first penetrated SDF, intersection.cu:135 */
            }
        }
    }
    free(de);
    sys_emit(&s, result);
    stat_enter(oldcat_);
    return nColl;
}

int orc_ne_intersection(const float *sites, int nSites, const float *T_ds, const float *T_sd,
                        const orc_model *src, const float *T_fm_src, const float *T_mf_src,
                        const orc_model *dst, const float *T_fm_dst, const float *T_mf_dst, int reduction,
                        int redDims, const float *dtheta_dalpha, const int32_t *mapping, float *result) {
    const int oldcat_ = stat_enter(ORC_STAT_INTERSECTION);
    const int fullDims = 6 + src->n_joints;
    const int dims = reduction ? redDims : fullDims;
    sys_acc s = sys_new(dims);
    float *de = (float *)malloc(sizeof(float) * (size_t)(fullDims + dims));
    float *Jr = de + fullDims;
    int nColl = 0;
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        const int srcFrame = src->sdf_frame[srcGrid];
        float vsm[4], vdm[4];
        se3_apply4(T_mf_src + 12 * srcFrame, vs, vsm);
        se3_apply4(T_ds, vsm, vdm);
        for (int dstGrid = 0; dstGrid < dst->n_sdfs; ++dstGrid) {
            const int dstFrame = dst->sdf_frame[dstGrid];
            float vd[4], vg[3];
            se3_apply4(T_fm_dst + 12 * dstFrame, vdm, vd);
            const orc_grid sdf = model_sdf(dst, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            const float residual = orc_grid_value_interp(&sdf, vg) * sdf.resolution;
            if (residual < 0) {
                float gf[3], gdm[3], gsm[3];
                orc_grid_gradient_interp(&sdf, vg, gf);
                se3_rotate3(T_mf_dst + 12 * dstFrame, gf, gdm);
                se3_rotate3(T_sd, gdm, gsm);
                orc_error_jacobian(src, T_fm_src, T_mf_src, vsm, srcFrame, gsm, 0, de);
                const float *J = de;
                if (reduction) { reduce_jacobian(reduction, 6, fullDims, redDims, de, dtheta_dalpha, mapping, Jr); J = Jr; }
                sys_add_squared(&s, residual, J);
                ++nColl;
                break;
            }
        }
    }
    free(de);
    sys_emit(&s, result);
    stat_enter(oldcat_);
    return nColl;
}

int orc_count_self_intersections(const float *sites, int nSites, const orc_model *m, const float *T_fm,
                                 const float *T_mf, const int32_t *pot) {
    int n = 0;
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        float vm[4];
        se3_apply4(T_mf + 12 * m->sdf_frame[srcGrid], vs, vm);
        for (int dstGrid = 0; dstGrid < m->n_sdfs; ++dstGrid) {
            if (!pot[dstGrid + srcGrid * m->n_sdfs]) continue;
            float vd[4], vg[3];
            se3_apply4(T_fm + 12 * m->sdf_frame[dstGrid], vm, vd);
            const orc_grid sdf = model_sdf(m, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            if (orc_grid_value_interp(&sdf, vg) * sdf.resolution < 0) { ++n; break; }
        }
    }
    return n;
}

int orc_count_intersections(const float *sites, int nSites, const float *T_ds, const orc_model *src,
                            const float *T_mf_src, const orc_model *dst, const float *T_fm_dst) {
    int n = 0;
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        float vsm[4], vdm[4];
        se3_apply4(T_mf_src + 12 * src->sdf_frame[srcGrid], vs, vsm);
        se3_apply4(T_ds, vsm, vdm);
        for (int dstGrid = 0; dstGrid < dst->n_sdfs; ++dstGrid) {
            float vd[4], vg[3];
            se3_apply4(T_fm_dst + 12 * dst->sdf_frame[dstGrid], vdm, vd);
            const orc_grid sdf = model_sdf(dst, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            if (orc_grid_value_interp(&sdf, vg) * sdf.resolution < 0) { ++n; break; }
        }
    }
    return n;
}

/* The kernels' optional debug outputs. modToObs.cu:47-49,65,74,82 (gpu_normEqnsModToObs; the Reduced / ParamMap kernels write
 * the same at :249-286, :350-387) and :145-147,168,178,186 (Truncated; truncationDist < 0 here = the plain kernels). Any map may be
 * NULL. dbgNorm: [height*width][4]. */
void orc_mod2obs_debug_maps(const float *pred, int width, int height, int modelNum, const float *T_gc, const orc_grid *obsSdf,
                            float truncationDist, int32_t *dbgDA, float *dbgErr, float *dbgNorm) {
    for (int index = 0; index < width * height; ++index) {
        if (dbgDA && modelNum == 0) dbgDA[index] = -1;
        if (dbgErr && modelNum == 0) dbgErr[index] = NAN;
        if (dbgNorm) dbgNorm[4 * index] = dbgNorm[4 * index + 1] = dbgNorm[4 * index + 2] = dbgNorm[4 * index + 3] = 0;
        const float *pc = pred + 4 * index;
        if (pc[2] == 0) continue;
        float pm[4], pg[3];
        se3_apply3(T_gc, pc, pm); pm[3] = 1;
        orc_grid_coords(obsSdf, pm, pg);
        if (!orc_grid_in_bounds_gradient_interp(obsSdf, pg)) continue;
        const float residual = orc_grid_value_interp(obsSdf, pg) * obsSdf->resolution;
        if (truncationDist >= 0 && (residual >= truncationDist || residual < 0)) continue;
        if (dbgErr) dbgErr[index] = residual;
        const int label = (int)round(pc[3]);
        if ((label >> 16) != modelNum) continue;
        if (dbgDA) dbgDA[index] = label;
        if (dbgNorm) {
            float gm[3];
            orc_grid_gradient_interp(obsSdf, pg, gm);
            dbgNorm[4 * index] = gm[0]; dbgNorm[4 * index + 1] = gm[1]; dbgNorm[4 * index + 2] = gm[2]; dbgNorm[4 * index + 3] = 1;
        }
    }
}

/* intersection.cu:303-313 (NaN everywhere), :131-133 (self term: a colliding site gets residual^2), :433-435 (pairwise term:
 * += residual^2); sites without a collision keep what they held */
void orc_init_debug_intersection_error(float *debugError, int nSites) {
    for (int i = 0; i < nSites; ++i) debugError[i] = NAN;
}
void orc_debug_self_intersection_error(const float *sites, int nSites, const orc_model *m, const float *T_fm, const float *T_mf,
                                       const int32_t *pot, float *debugError) {
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        float vm[4];
        se3_apply4(T_mf + 12 * m->sdf_frame[srcGrid], vs, vm);
        for (int dstGrid = 0; dstGrid < m->n_sdfs; ++dstGrid) {
            if (!pot[dstGrid + srcGrid * m->n_sdfs]) continue;
            float vd[4], vg[3];
            se3_apply4(T_fm + 12 * m->sdf_frame[dstGrid], vm, vd);
            const orc_grid sdf = model_sdf(m, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            const float residual = orc_grid_value_interp(&sdf, vg) * sdf.resolution;
            if (residual < 0) { debugError[index] = residual * residual; break; }
        }
    }
}
void orc_debug_intersection_error(const float *sites, int nSites, const float *T_ds, const orc_model *src, const float *T_mf_src,
                                  const orc_model *dst, const float *T_fm_dst, float *debugError) {
    for (int index = 0; index < nSites; ++index) {
        float vs[4] = {sites[4 * index], sites[4 * index + 1], sites[4 * index + 2], 1};
        const int srcGrid = (int)round(sites[4 * index + 3]);
        float vsm[4], vdm[4];
        se3_apply4(T_mf_src + 12 * src->sdf_frame[srcGrid], vs, vsm);
        se3_apply4(T_ds, vsm, vdm);
        for (int dstGrid = 0; dstGrid < dst->n_sdfs; ++dstGrid) {
            float vd[4], vg[3];
            se3_apply4(T_fm_dst + 12 * dst->sdf_frame[dstGrid], vdm, vd);
            const orc_grid sdf = model_sdf(dst, dstGrid);
            orc_grid_coords(&sdf, vd, vg);
            if (!orc_grid_in_bounds_gradient_interp(&sdf, vg)) continue;
            const float residual = orc_grid_value_interp(&sdf, vg) * sdf.resolution;
            if (residual < 0) { debugError[index] += residual * residual; break; }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * assembly + solve
 * ---------------------------------------------------------------------------------------------- */
float orc_unpack(float *eJ, float *JTJ, int ld, const float *sys, float multiplier, int dims) {
    const float *sys_eJ = sys, *sys_JTJ = sys + dims, *sys_e = sys + dims + jtj_size(dims);
    for (int j = 0; j < dims; ++j) {
        eJ[j] += multiplier * sys_eJ[j];
        for (int i = 0; i <= j; ++i) JTJ[i * ld + j] += multiplier * sys_JTJ[((j * (j + 1)) >> 1) + i];
    }
    return multiplier * (*sys_e);
}

void orc_solve_ldlt(const float *Ain, const float *b, int n, float *x) {
    /* LDL^T with symmetric diagonal pivoting (largest |diagonal| first), lower-triangular in-place, fp32
     * like Eigen's MatrixXf::ldlt(); then solve P^T L D L^T P x = b. */
    float *A = (float *)malloc(sizeof(float) * (size_t)n * n);
    int *perm = (int *)malloc(sizeof(int) * (size_t)n);
    float *y = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(A, Ain, sizeof(float) * (size_t)n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k; float best = fabsf(A[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (fabsf(A[i * n + i]) > best) { best = fabsf(A[i * n + i]); piv = i; }
        if (piv != k) { /* symmetric row/column swap */
            for (int j = 0; j < n; ++j) { float t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            for (int i = 0; i < n; ++i) { float t = A[i * n + k]; A[i * n + k] = A[i * n + piv]; A[i * n + piv] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        const float d = A[k * n + k];
        if (d == 0.0f) continue; /* Eigen leaves the remaining columns untouched when the pivot is exactly zero */
        for (int i = k + 1; i < n; ++i) A[i * n + k] /= d;
        for (int i = k + 1; i < n; ++i) {
            const float lik = A[i * n + k];
            for (int j = k + 1; j <= i; ++j) {
                A[i * n + j] -= lik * d * A[j * n + k];
                A[j * n + i] = A[i * n + j];
            }
        }
    }
    for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
    for (int i = 0; i < n; ++i) { float s = y[i]; for (int j = 0; j < i; ++j) s -= A[i * n + j] * y[j]; y[i] = s; }
    for (int i = 0; i < n; ++i) {
        const float d = A[i * n + i];
        /* Eigen's pseudo-inverse of D: tiny pivots give 0 (LDLT::_solve_impl tolerance = 1/max float) */
        y[i] = (fabsf(d) > FLT_MIN) ? y[i] / d : 0.0f;
    }
    for (int i = n - 1; i >= 0; --i) { float s = y[i]; for (int j = n - 1; j > i; --j) s -= A[j * n + i] * y[j]; y[i] = s; }
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    free(A); free(perm); free(y);
}

void orc_solve_upper(const float *A, const float *b, int n, float *x) {
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int i = n - 1; i >= 0; --i) {
        float tmp = x[i];
        if (tmp != 0.0f) {
            x[i] = tmp / A[i * n + i];
            tmp = x[i];
            for (int r = 0; r < i; ++r) {
                const float a = A[r * n + i];
                if (a != 0.0f) x[r] -= tmp * a;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * projectToSdfSurface (sdf.cpp:6-23): gradient steps of the signed distance towards the zero level set, in grid
 * coordinates. `abs` there is the float overload (sdf.h pulls in <math.h> and <stdlib.h> through SE3.h / grid_3d.h).
 * normalize() is helper_math's v * (1 / sqrtf(dot(v, v))) on the host.
 * ---------------------------------------------------------------------------------------------- */
void orc_project_to_sdf_surface(const orc_grid *sdf, float *pg, float threshold, int maxIters) {
    for (int iter = 0; iter < maxIters; ++iter) {
        if (!orc_grid_in_bounds_gradient_interp(sdf, pg)) return;
        const float dist = orc_grid_value_interp(sdf, pg);
        if (fabsf(dist) < threshold) return;
        float grad[3];
        orc_grid_gradient_interp(sdf, pg, grad);
        const float invLen = 1.0f / sqrtf(dot3(grad, grad));
        for (int k = 0; k < 3; ++k) pg[k] = pg[k] - dist * (grad[k] * invLen);
    }
}

/* ContactPrior::updatePriorParams (contact_prior.cpp:207-221): the contact point takes its three entries of the parameter
 * update and is projected back onto the surface of its own (source) SDF; threshold 1e-9, the default 100 iterations. */
void orc_contact_prior_update_params(orc_contact_prior *p, const float *update, const orc_model *const *models) {
    if (!p->updateContact) return;
    for (int k = 0; k < 3; ++k) p->contactPoint[k] += update[k];
    const orc_grid sdf = model_sdf(models[p->srcModel], p->srcSdf);
    float g[3];
    orc_grid_coords(&sdf, p->contactPoint, g);
    orc_project_to_sdf_surface(&sdf, g, 1e-9f, 100);
    for (int k = 0; k < 3; ++k) p->contactPoint[k] = sdf.resolution * g[k] + sdf.offset[k];      /* getWorldCoords, grid_3d.h:53-55 */
}

/* ------------------------------------------------------------------------------------------------
 * contact prior (contact_prior.cpp:8-205)
 * ---------------------------------------------------------------------------------------------- */
static float contact_prior_contribution_(const orc_contact_prior *p, const orc_model *const *models,
                                         const float *const *T_mcs, const float *const *T_cms,
                                         const float *const *T_fms, const float *const *T_mfs,
                                         const int32_t *modelOffsets, const int32_t *modelDims,
                                         int priorParamOffset, float *JTJ, int sys, float *JTe,
                                         int32_t *dstSdfOut);
float orc_contact_prior_contribution(const orc_contact_prior *p, const orc_model *const *models,
                                     const float *const *T_mcs, const float *const *T_cms,
                                     const float *const *T_fms, const float *const *T_mfs,
                                     const int32_t *modelOffsets, const int32_t *modelDims,
                                     int priorParamOffset, float *JTJ, int sys, float *JTe,
                                     int32_t *dstSdfOut) {
    const int oldcat = stat_enter(ORC_STAT_PRIOR);
    const float e = contact_prior_contribution_(p, models, T_mcs, T_cms, T_fms, T_mfs, modelOffsets, modelDims, priorParamOffset,
                                                JTJ, sys, JTe, dstSdfOut);
    stat_enter(oldcat);
    return e;
}
static float contact_prior_contribution_(const orc_contact_prior *p, const orc_model *const *models,
                                         const float *const *T_mcs, const float *const *T_cms,
                                         const float *const *T_fms, const float *const *T_mfs,
                                         const int32_t *modelOffsets, const int32_t *modelDims,
                                         int priorParamOffset, float *JTJ, int sys, float *JTe,
                                         int32_t *dstSdfOut) {
    const int sign = p->invert ? -1 : 1;
    float error = sign * p->weight * p->contactThreshold;
    if (dstSdfOut) *dstSdfOut = -1;
    if (p->weight == 0) return error;
    const orc_model *src = models[p->srcModel], *dst = models[p->dstModel];
    const int srcOffset = modelOffsets[p->srcModel], dstOffset = modelOffsets[p->dstModel];
    const int srcDims = modelDims[p->srcModel], dstDims = modelDims[p->dstModel];
    const int srcFrame = src->sdf_frame[p->srcSdf];
    const float csf[4] = {p->contactPoint[0], p->contactPoint[1], p->contactPoint[2], 1};
    float csm[4], cc[4], cdm[4];
    se3_apply4(T_mfs[p->srcModel] + 12 * srcFrame, csf, csm);
    se3_apply4(T_cms[p->srcModel], csm, cc);
    se3_apply4(T_mcs[p->dstModel], cc, cdm);
    /* nearest destination SDF by raw voxel-unit value (contact_prior.cpp:41-54). The reference loop runs
     * dsidx <= getNumSdfs(), reading one element past the end (undefined behaviour); restated with '<'. */
    int dstSdfNum = -1;
    float min_dsdf = FLT_MAX;
    for (int ds = 0; ds < dst->n_sdfs; ++ds) {
        float df[4], dfg[3];
        se3_apply4(T_fms[p->dstModel] + 12 * dst->sdf_frame[ds], cdm, df);
        const orc_grid g = model_sdf(dst, ds);
        orc_grid_coords(&g, df, dfg);
        if (!orc_grid_in_bounds_interp(&g, dfg)) continue;
        const float dsdf = orc_grid_value_interp(&g, dfg);
        if (dsdf < min_dsdf) { min_dsdf = dsdf; dstSdfNum = ds; }
    }
    if (dstSdfOut) *dstSdfOut = dstSdfNum;
    if (dstSdfNum < 0) return error;
    const int dstFrame = dst->sdf_frame[dstSdfNum];
    float cdf[4], cg[3];
    se3_apply4(T_fms[p->dstModel] + 12 * dstFrame, cdm, cdf);
    const orc_grid dsdf = model_sdf(dst, dstSdfNum);
    orc_grid_coords(&dsdf, cdf, cg);
    if (!orc_grid_in_bounds_gradient_interp(&dsdf, cg)) return error;
    float errContact = orc_grid_value_interp(&dsdf, cg) * dsdf.resolution;
    if (fabsf(errContact) > p->contactThreshold) return error;
    if (fabsf(errContact) < 1e-4) return 0.0f;

    float g_df[3], g_dm[3], g_c[3], g_sm[3], g_sf[3];
    orc_grid_gradient_interp(&dsdf, cg, g_df);
    se3_rotate3(T_mfs[p->dstModel] + 12 * dstFrame, g_df, g_dm);
    se3_rotate3(T_cms[p->dstModel], g_dm, g_c);
    se3_rotate3(T_mcs[p->srcModel], g_c, g_sm);
    se3_rotate3(T_fms[p->srcModel] + 12 * srcFrame, g_sm, g_sf);

    if (p->invert && errContact > 0) {
        const float invLen = 1.0f / sqrtf(dot3(g_sf, g_sf));
        const float n[3] = {g_sf[0] * invLen, g_sf[1] * invLen, g_sf[2] * invLen};
        const float cos_theta = dot3(p->normal, n);
        const float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
        if (fabsf(errContact * sin_theta) > 0.01f) return error;
    }
    if (p->invert) errContact *= -1.f;
    error = sign * p->weight * fabsf(errContact);
    if (!(p->updateContact || p->updateSrc || p->updateDst)) return error;

    const int sub = srcDims + dstDims + 3;
    float *subJ = (float *)calloc((size_t)sub, sizeof(float));
    if (p->updateContact) { subJ[srcDims + dstDims] = g_sf[0]; subJ[srcDims + dstDims + 1] = g_sf[1]; subJ[srcDims + dstDims + 2] = g_sf[2]; }
    if (p->updateDst) {
        float *J3 = (float *)malloc(sizeof(float) * 3 * (size_t)(6 + dst->n_joints));
        orc_model_jacobian_3d(dst, T_fms[p->dstModel], T_mfs[p->dstModel], cdm, dstFrame, J3);
        for (int i = 0; i < dstDims; ++i) subJ[srcDims + i] = -dot3(g_dm, J3 + 3 * i);
        free(J3);
    }
    if (p->updateSrc) {
        float *J3 = (float *)malloc(sizeof(float) * 3 * (size_t)(6 + src->n_joints));
        orc_model_jacobian_3d(src, T_fms[p->srcModel], T_mfs[p->srcModel], csm, srcFrame, J3);
        for (int i = 0; i < srcDims; ++i) subJ[i] = dot3(g_sm, J3 + 3 * i);
        free(J3);
    }
    const float we = p->weight * errContact;
#define ORC_SUBJTJ(a, b) (p->weight * subJ[a] * subJ[b])
    for (int i = 0; i < srcDims; ++i) JTe[srcOffset + i] += we * subJ[i];
    for (int i = 0; i < dstDims; ++i) JTe[dstOffset + i] += we * subJ[srcDims + i];
    for (int i = 0; i < 3; ++i) JTe[priorParamOffset + i] += we * subJ[srcDims + dstDims + i];
    for (int i = 0; i < srcDims; ++i) for (int j = i; j < srcDims; ++j) {
        const float val = ORC_SUBJTJ(i, j); if (val == 0) continue;
        JTJ[(srcOffset + i) * sys + srcOffset + j] += val;
    }
    for (int i = 0; i < dstDims; ++i) for (int j = i; j < dstDims; ++j) {
        const float val = ORC_SUBJTJ(srcDims + i, srcDims + j); if (val == 0) continue;
        JTJ[(dstOffset + i) * sys + dstOffset + j] += val;
    }
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) {
        const float val = ORC_SUBJTJ(srcDims + dstDims + i, srcDims + dstDims + j); if (val == 0) continue;
        JTJ[(priorParamOffset + i) * sys + priorParamOffset + j] += val;
    }
    /* contact_prior.cpp:159-170 mutates subJTJ's contact diagonal after it was scattered: no effect. */
    for (int i = 0; i < srcDims; ++i) for (int j = 0; j < 3; ++j) {
        const float val = ORC_SUBJTJ(i, srcDims + dstDims + j); if (val == 0) continue;
        JTJ[(srcOffset + i) * sys + priorParamOffset + j] += val;
    }
    for (int i = 0; i < dstDims; ++i) for (int j = 0; j < 3; ++j) {
        const float val = ORC_SUBJTJ(srcDims + i, srcDims + dstDims + j); if (val == 0) continue;
        JTJ[(dstOffset + i) * sys + priorParamOffset + j] += val;
    }
    if (srcOffset < dstOffset) {
        for (int i = 0; i < srcDims; ++i) for (int j = 0; j < dstDims; ++j) {
            const float val = ORC_SUBJTJ(i, srcDims + j); if (val == 0) continue;
            JTJ[(srcOffset + i) * sys + dstOffset + j] += val;
        }
    } else {
        for (int i = 0; i < dstDims; ++i) for (int j = 0; j < srcDims; ++j) {
            const float val = ORC_SUBJTJ(srcDims + i, j); if (val == 0) continue;
            JTJ[(dstOffset + i) * sys + srcOffset + j] += val;
        }
    }
#undef ORC_SUBJTJ
    free(subJ);
    return error;
}

/* ------------------------------------------------------------------------------------------------
 * Point3D3DPrior / Point2D3DPrior::computeContribution (point_3d_3d_prior.cpp:6-69, point_2d_3d_prior.cpp:6-83)
 * ---------------------------------------------------------------------------------------------- */
void orc_point_prior_contribution(const orc_point_prior *p, const orc_model *const *models, const float *const *T_mcs,
                                  const float *const *T_cms, const float *const *T_fms, const float *const *T_mfs,
                                  const int32_t *modelOffsets, const int32_t *modelDims, float *JTJ, int sys, float *JTe) {
    if (p->weight == 0) return;
    const orc_model *src = models[p->srcModel];
    const int srcOffset = modelOffsets[p->srcModel], srcDims = modelDims[p->srcModel];
    const float *T_mf = T_mfs[p->srcModel] + 12 * p->srcFrame;
    float T_fc[12];
    orc_se3_mul(T_cms[p->srcModel], T_mf, T_fc);              /* getTransformFrameToCamera = T_cm * T_mf */
    const float pf[4] = {p->point_f[0], p->point_f[1], p->point_f[2], 1};
    float pc[4], pm[4];
    se3_apply4(T_fc, pf, pc);
    se3_apply4(T_mf, pf, pm);
    const int rows = p->kind == 0 ? 3 : 2;
    float subE[3] = {0, 0, 0};
    float g[3][3];       /* row k of subJ is dot(g[k], J3D_i) with g in the model frame */
    if (p->kind == 0) {
        const float dc[3] = {pc[0] - p->point_c[0], pc[1] - p->point_c[1], pc[2] - p->point_c[2]};
        const float dist = sqrtf(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
        if (dist == 0) return;
        se3_rotate3(T_mcs[p->srcModel], dc, subE);              /* diff_m */
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) g[k][a] = (k == a) ? 1.0f : 0.0f;
    } else {
        const float fx = p->focalLength[0], fy = p->focalLength[1];
        const float px = pc[0] * fx / pc[2] + p->principalPoint[0];
        const float py = pc[1] * fy / pc[2] + p->principalPoint[1];
        const float dx = px - p->point_c[0], dy = py - p->point_c[1];
        if (sqrtf(dx * dx + dy * dy) == 0) return;
        subE[0] = dx; subE[1] = dy;
        /* subJ(k,i) = dot(gc_k, R_cm J3D_i) = dot(R_mc gc_k, J3D_i) */
        const float gc0[3] = {fx / pc[2], 0, -(pc[0]) * fx / (pc[2] * pc[2])};
        const float gc1[3] = {0, fy / pc[2], -(pc[1]) * fy / (pc[2] * pc[2])};
        se3_rotate3(T_mcs[p->srcModel], gc0, g[0]);
        se3_rotate3(T_mcs[p->srcModel], gc1, g[1]);
    }
    float *J3 = (float *)malloc(sizeof(float) * 3 * (size_t)(6 + src->n_joints));
    float *subJ = (float *)malloc(sizeof(float) * 3 * (size_t)srcDims);
    orc_model_jacobian_3d(src, T_fms[p->srcModel], T_mfs[p->srcModel], pm, p->srcFrame, J3);
    for (int k = 0; k < rows; ++k)
        for (int i = 0; i < srcDims; ++i) subJ[k * srcDims + i] = dot3(g[k], J3 + 3 * i);
    for (int i = 0; i < srcDims; ++i) {
        float v = 0;
        for (int k = 0; k < rows; ++k) v += (p->weight * subJ[k * srcDims + i]) * subE[k];
        JTe[srcOffset + i] += v;
    }
    for (int i = 0; i < srcDims; ++i) for (int j = i; j < srcDims; ++j) {
        float v = 0;
        for (int k = 0; k < rows; ++k) v += (p->weight * subJ[k * srcDims + i]) * subJ[k * srcDims + j];
        if (v == 0) continue;
        JTJ[(srcOffset + i) * sys + srcOffset + j] += v;
    }
    free(J3); free(subJ);
}
