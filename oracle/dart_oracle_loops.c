/*
 * dart_oracle_loops.c -- CPU ORACLE (test infrastructure, NOT product code): the optimizer's host loops
 * (optimizer.cpp) restated over the term functions of dart_oracle.c. Also the OpenMP-parallel CPU baseline
 * that bench.py times ("port"), which calls exactly these functions.
 */
#include "dart_oracle.h"
#include "dart_oracle_loops.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int jtj_size(int d) { return (d * (d + 1)) >> 1; }
static int sys_len(int d) { return d + jtj_size(d) + 1; }

/* Optimizer::optimizePose, optimizer.cpp:423-522 (NullReduction poses). */
void orc_optimize_pose(const orc_model *m, float *T_mc, float *joints, const float *vert, const float *norm,
                       int width, int height, const orc_opt_options *o, const float *cloud, int nCloud,
                       const int32_t *potential, float *obsSdfData, float *iterInfo, float *firstSystem) {
    const int dims = 6 + m->n_joints;
    const int nF = m->n_frames;
    float *T_mf = (float *)malloc(sizeof(float) * 12 * (size_t)nF);
    float *T_fm = (float *)malloc(sizeof(float) * 12 * (size_t)nF);
    float *JTJ = (float *)malloc(sizeof(float) * (size_t)dims * dims);
    float *A = (float *)malloc(sizeof(float) * (size_t)dims * dims);
    float *eJ = (float *)malloc(sizeof(float) * (size_t)dims);
    float *dalpha = (float *)malloc(sizeof(float) * (size_t)dims);
    float *sys = (float *)malloc(sizeof(float) * (size_t)sys_len(dims));
    orc_da_point *pts = (orc_da_point *)malloc(sizeof(orc_da_point) * (size_t)width * height);
    float *pred = (float *)calloc((size_t)o->predWidth * o->predHeight * 4, sizeof(float));
    const float lmLambda = o->regularizationScaled;
    float T_obsSdf_c[12];
    orc_grid obs;
    obs.dim[0] = m->obs_sdf_dim[0]; obs.dim[1] = m->obs_sdf_dim[1]; obs.dim[2] = m->obs_sdf_dim[2];
    obs.offset[0] = m->obs_sdf_offset[0]; obs.offset[1] = m->obs_sdf_offset[1]; obs.offset[2] = m->obs_sdf_offset[2];
    obs.resolution = m->obs_sdf_resolution;
    obs.data = obsSdfData;

    if (o->lambdaModToObs > 0) { /* generateObsSdf, optimizer.cpp:168-219 */
        float T_cm[12];
        const size_t n = (size_t)obs.dim[0] * obs.dim[1] * obs.dim[2];
        float *tmp = (float *)malloc(sizeof(float) * n);
        orc_se3_invert(T_mc, T_cm);
        orc_splat_obs_sdf(vert, width, height, T_cm, &obs, o->focalLength, obsSdfData);
        orc_distance_transform_3d(obsSdfData, tmp, obs.dim[0], obs.dim[1], obs.dim[2], 1);
        memcpy(obsSdfData, tmp, sizeof(float) * n);
        free(tmp);
        memcpy(T_obsSdf_c, T_mc, sizeof(float) * 12);
    }

    for (int it = 0; it < o->numIterations; ++it) {
        orc_fk_set_pose(m, joints, T_mf, T_fm); /* pose.projectReducedToFull(); model.setPose(pose) */
        if (o->lambdaModToObs > 0)
            orc_raycast_prediction(o->predFl[0], o->predFl[1], (float)(o->predWidth / 2), (float)(o->predHeight / 2),
                                   o->predWidth, o->predHeight, 0, m, T_mc, T_fm, T_mf, 0.0f, pred);
        memset(JTJ, 0, sizeof(float) * (size_t)dims * dims);
        memset(eJ, 0, sizeof(float) * (size_t)dims);
        float eO2M = 0, eM2O = 0, eInt = 0;
        int nAssoc = 0, nPred = 0;
        if (o->lambdaObsToMod > 0) {
            nAssoc = orc_assoc_single(vert, norm, width, height, m, T_mc, T_fm, o->distThreshold, o->normThreshold,
                                      o->planeOffset, o->planeNormal, pts, NULL, NULL, NULL);
            orc_ne_obs2mod(vert, m, T_mc, T_fm, T_mf, pts, nAssoc, o->huberDelta, 0, dims, NULL, NULL, sys);
            if (it == 0 && firstSystem) memcpy(firstSystem, sys, sizeof(float) * (size_t)sys_len(dims));
            eO2M = orc_unpack(eJ, JTJ, dims, sys, o->lambdaObsToMod, dims);
        }
        if (o->lambdaModToObs > 0) {
            nPred = orc_ne_mod2obs(pred, o->predWidth, o->predHeight, m->model_id, m, T_obsSdf_c, T_fm, T_mf, &obs, -1.0f, 0,
                                   dims, NULL, NULL, sys);
            if (it == 0 && firstSystem) memcpy(firstSystem + sys_len(dims), sys, sizeof(float) * (size_t)sys_len(dims));
            eM2O = orc_unpack(eJ, JTJ, dims, sys, o->lambdaModToObs, dims);
        }
        if (o->lambdaIntersection > 0 && nCloud > 0) { /* computeSelfIntersectionContribution, :332-384 */
            const int ad = dims - 6;
            orc_ne_self_intersection(cloud, nCloud, m, T_fm, T_mf, potential, 0, dims, NULL, NULL, sys);
            if (it == 0 && firstSystem) memcpy(firstSystem + 2 * sys_len(dims), sys, sizeof(float) * (size_t)sys_len(ad));
            if (ad > 0) {
                float *eJt = (float *)calloc((size_t)ad, sizeof(float));
                float *Jt = (float *)calloc((size_t)ad * ad, sizeof(float));
                eInt = orc_unpack(eJt, Jt, ad, sys, o->lambdaIntersection, ad);
                for (int i = 0; i < ad; ++i) {
                    eJ[6 + i] += eJt[i];
                    for (int j = 0; j < ad; ++j) JTJ[(6 + i) * dims + 6 + j] += Jt[i * ad + j];
                }
                free(eJt); free(Jt);
            }
        }
        if (iterInfo) {
            float *r = iterInfo + 8 * it;
            r[0] = eO2M; r[1] = eM2O; r[2] = eInt; r[3] = (float)nAssoc; r[4] = (float)nPred; r[5] = r[6] = r[7] = 0;
        }
        for (int i = 0; i < dims; ++i) for (int j = 0; j < i; ++j) JTJ[i * dims + j] = JTJ[j * dims + i];
        for (int i = 0; i < dims; ++i) JTJ[i * dims + i] += o->regularization * 1.0f;
        memcpy(A, JTJ, sizeof(float) * (size_t)dims * dims);
        for (int i = 0; i < dims; ++i) A[i * dims + i] += lmLambda * A[i * dims + i];
        orc_solve_ldlt(A, eJ, dims, dalpha);
        for (int i = 0; i < dims; ++i) dalpha[i] = -dalpha[i];
        float dT[12];
        orc_se3_exp(dalpha, dT);
        orc_se3_mul(dT, T_mc, T_mc);
        for (int i = 0; i < m->n_joints; ++i) {
            const float v = joints[i] + dalpha[i + 6];
            joints[i] = fminf(fmaxf(m->joint_limits[2 * i], v), m->joint_limits[2 * i + 1]);
        }
    }
    free(T_mf); free(T_fm); free(JTJ); free(A); free(eJ); free(dalpha); free(sys); free(pts); free(pred);
}
