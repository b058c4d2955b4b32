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

/* Test aid (no reference counterpart): resume a run mid-way. The reference builds the observation SDF (and freezes
 * T_obsSdf_c) from the pose an optimizePose(s) call STARTS with (optimizer.cpp:446-448, 574-580); to continue a trajectory
 * from an intermediate pose exactly as the original call would have, the next orc_optimize_pose / orc_optimize_poses call
 * takes the observation-SDF pose(s) from here ([nModels][12], camera->model) instead of from its own start poses. NULL (the
 * default) restores the reference behaviour. */
static const float *g_obsStart = NULL;
void orc_set_obs_sdf_start(const float *T_mcs) { g_obsStart = T_mcs; }

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
        const float *T_start = g_obsStart ? g_obsStart : T_mc;
        orc_se3_invert(T_start, T_cm);
        orc_splat_obs_sdf(vert, width, height, T_cm, &obs, o->focalLength, obsSdfData);
        orc_distance_transform_3d(obsSdfData, tmp, obs.dim[0], obs.dim[1], obs.dim[2], 1);
        memcpy(obsSdfData, tmp, sizeof(float) * n);
        free(tmp);
        memcpy(T_obsSdf_c, T_start, sizeof(float) * 12);
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

/* ------------------------------------------------------------------------------------------------
 * OptimizerNoObs::optimizePoses / getError (optimizer.cpp:783-1076)
 * ---------------------------------------------------------------------------------------------- */
/* Pose reductions of the NoObs path (grasp_analyzer.cpp:393-420: Barrett's LinearPoseReduction(8 -> 4)); [nModels] or NULL.
 * With a reduction the joints[m] arrays of orc_optimize_poses_noobs / orc_get_error_noobs carry the REDUCED articulation. */
static const orc_reduction *g_noobsRed = NULL;
void orc_noobs_set_reductions(const orc_reduction *red) { g_noobsRed = red; }
static void project_reduced(const orc_model *m, const orc_reduction *r, const float *art, float *full);
static int noobs_red_type(int m) { return g_noobsRed ? g_noobsRed[m].type : 0; }

/* Offsets of the priors' 3 parameters behind the model blocks (optimizer.cpp:590-593, 829-832: every ContactPrior owns
 * three, getNumPriorParams). Frozen contact points never receive a non-zero in their rows or columns, so they share one
 * all-zero slot at the end here instead of three unknowns each. Returns the system size. */
static int prior_param_offsets(const orc_contact_prior *priors, int nPriors, int D, int *off) {
    int sys = D, anyFrozen = 0;
    for (int p = 0; p < nPriors; ++p) {
        if (priors[p].updateContact) { off[p] = sys; sys += 3; }
        else { off[p] = -1; anyFrozen = 1; }
    }
    if (anyFrozen || nPriors == 0) { for (int p = 0; p < nPriors; ++p) if (off[p] < 0) off[p] = sys; sys += 3; }
    return sys;
}
/* where the loops leave the contact points of their priors after the last iteration ([nPriors][3]; the reference mutates the
 * prior objects, contact_prior.cpp:212-219), or NULL */
static float *g_contactOut = NULL;
void orc_set_contact_out(float *out) { g_contactOut = out; }
/* Tracker / TrackerNoObs::getDampingMatrix (tracker.h:122, tracker_no_obs.h:73): one [dims][dims] row-major matrix per model
 * (NULL entries or a NULL list = zero, the trackers' initial value), used by both loops below until reset */
static const float *const *g_damping = NULL;
void orc_set_damping(const float *const *mats) { g_damping = mats; }
static orc_contact_prior *priors_copy(const orc_contact_prior *priors, int nPriors) {
    orc_contact_prior *c = (orc_contact_prior *)malloc(sizeof(orc_contact_prior) * (size_t)(nPriors > 0 ? nPriors : 1));
    if (nPriors > 0) memcpy(c, priors, sizeof(orc_contact_prior) * (size_t)nPriors);
    return c;
}
static void contacts_out(const orc_contact_prior *priors, int nPriors) {
    if (!g_contactOut) return;
    for (int p = 0; p < nPriors; ++p) for (int k = 0; k < 3; ++k) g_contactOut[3 * p + k] = priors[p].contactPoint[k];
}

typedef struct {
    int n, D, sys, nPriors;
    float *full[ORC_MAX_MODELS];     /* full joint vectors (projection of the reduced articulation) */
    int offs[ORC_MAX_MODELS], dims[ORC_MAX_MODELS];
    float *T_mf[ORC_MAX_MODELS], *T_fm[ORC_MAX_MODELS], *T_cm[ORC_MAX_MODELS];
    const float *cT_mc[ORC_MAX_MODELS], *cT_cm[ORC_MAX_MODELS], *cT_fm[ORC_MAX_MODELS], *cT_mf[ORC_MAX_MODELS];
    int *priorOff;
} noobs_ctx;

static void noobs_init(noobs_ctx *c, const orc_model *const *models, int n, const orc_contact_prior *priors, int nPriors) {
    c->n = n; c->D = 0; c->nPriors = nPriors;
    for (int m = 0; m < n; ++m) {
        /* getReducedDimensions(), optimizer.cpp:821-823 */
        c->offs[m] = c->D; c->dims[m] = 6 + (noobs_red_type(m) ? g_noobsRed[m].redArt : models[m]->n_joints); c->D += c->dims[m];
        c->full[m] = (float *)calloc((size_t)models[m]->n_joints + 1, sizeof(float));
        c->T_mf[m] = (float *)malloc(sizeof(float) * 12 * (size_t)models[m]->n_frames);
        c->T_fm[m] = (float *)malloc(sizeof(float) * 12 * (size_t)models[m]->n_frames);
        c->T_cm[m] = (float *)malloc(sizeof(float) * 12);
    }
    c->priorOff = (int *)malloc(sizeof(int) * (size_t)(nPriors > 0 ? nPriors : 1));
    c->sys = prior_param_offsets(priors, nPriors, c->D, c->priorOff);
}
static void noobs_free(noobs_ctx *c) {
    for (int m = 0; m < c->n; ++m) { free(c->T_mf[m]); free(c->T_fm[m]); free(c->T_cm[m]); free(c->full[m]); }
    free(c->priorOff);
}
static void noobs_set_poses(noobs_ctx *c, const orc_model *const *models, float *const *T_mcs, float *const *joints) {
    for (int m = 0; m < c->n; ++m) {
        /* poses[m].projectReducedToFull(); models[m]->setPose(poses[m])  (optimizer.cpp:841-846) */
        project_reduced(models[m], g_noobsRed ? &g_noobsRed[m] : NULL, joints[m], c->full[m]);
        orc_fk_set_pose(models[m], c->full[m], c->T_mf[m], c->T_fm[m]);
        orc_se3_invert(T_mcs[m], c->T_cm[m]);
        c->cT_mc[m] = T_mcs[m]; c->cT_cm[m] = c->T_cm[m]; c->cT_fm[m] = c->T_fm[m]; c->cT_mf[m] = c->T_mf[m];
    }
}

/* intersection terms of model m into (eJ, JTJ) [dims x dims]; returns the sum of the lambda-scaled errors */
static float noobs_intersection_terms(noobs_ctx *c, const orc_model *const *models, float *const *T_mcs, int m,
                                      const orc_noobs_options *o, const float *const *clouds, const int32_t *nClouds,
                                      const int32_t *const *potentials, float *eJ, float *JTJ, float *sys) {
    const int n = c->n, dims = c->dims[m], ad = dims - 6;
    const int redType = noobs_red_type(m);
    const float *A = redType == 1 ? g_noobsRed[m].A : NULL;
    const int32_t *map = redType == 2 ? g_noobsRed[m].mapping : NULL;
    float total = 0.0f;
    if (o->lambdaIntersection[m + m * n] > 0 && ad > 0) { /* computeSelfIntersectionContribution, optimizer.cpp:332-384 */
        orc_ne_self_intersection(clouds[m], nClouds[m], models[m], c->T_fm[m], c->T_mf[m], potentials[m], redType, dims, A, map, sys);
        if (ad > 0) {
            float *eJt = (float *)calloc((size_t)ad, sizeof(float));
            float *Jt = (float *)calloc((size_t)ad * ad, sizeof(float));
            total += orc_unpack(eJt, Jt, ad, sys, o->lambdaIntersection[m + m * n], ad);
            for (int i = 0; i < ad; ++i) {
                eJ[6 + i] += eJt[i];
                for (int j = 0; j < ad; ++j) JTJ[(6 + i) * dims + 6 + j] += Jt[i * ad + j];
            }
            free(eJt); free(Jt);
        }
    }
    for (int d = 0; d < n; ++d) { /* computeIntersectionContribution, optimizer.cpp:386-421 */
        if (d == m || !(o->lambdaIntersection[m + d * n] > 0)) continue;
        float T_ds[12], T_sd[12];
        orc_se3_mul(T_mcs[d], c->T_cm[m], T_ds);
        orc_se3_invert(T_ds, T_sd);
        orc_ne_intersection(clouds[m], nClouds[m], T_ds, T_sd, models[m], c->T_fm[m], c->T_mf[m], models[d], c->T_fm[d], c->T_mf[d],
                            redType, dims, A, map, sys);
        total += orc_unpack(eJ, JTJ, dims, sys, o->lambdaIntersection[m + d * n], dims);
    }
    return total;
}

static const orc_point_prior *g_pointPriors = NULL;
static int g_nPointPriors = 0;
void orc_noobs_set_point_priors(const orc_point_prior *priors, int n) { g_pointPriors = n > 0 ? priors : NULL; g_nPointPriors = n > 0 ? n : 0; }

float orc_get_error_noobs(const orc_model *const *models, float *const *T_mcs, float *const *joints,
                          const orc_noobs_options *o, const float *const *clouds, const int32_t *nClouds,
                          const int32_t *const *potentials, const orc_contact_prior *priors, int nPriors) {
    noobs_ctx c;
    noobs_init(&c, models, o->nModels, priors, nPriors);
    noobs_set_poses(&c, models, T_mcs, joints);
    float error = 0.0f;
    float *JTJ = (float *)calloc((size_t)c.sys * c.sys, sizeof(float));
    float *JTe = (float *)calloc((size_t)c.sys, sizeof(float));
    for (int m = 0; m < c.n; ++m) {
        const int dims = c.dims[m];
        float *eJ = (float *)calloc((size_t)dims, sizeof(float));
        float *J = (float *)calloc((size_t)dims * dims, sizeof(float));
        float *sys = (float *)malloc(sizeof(float) * (size_t)sys_len(6 + models[m]->n_joints));
        /* getError adds every term's lambda-scaled e (optimizer.cpp:1047,1055) */
        error += noobs_intersection_terms(&c, models, T_mcs, m, o, clouds, nClouds, potentials, eJ, J, sys);
        free(eJ); free(J); free(sys);
    }
    float pe = 0.0f;
    for (int p = 0; p < nPriors; ++p)
        pe += orc_contact_prior_contribution(&priors[p], models, c.cT_mc, c.cT_cm, c.cT_fm, c.cT_mf, c.offs, c.dims, c.priorOff[p],
                                             JTJ, c.sys, JTe, NULL);
    if (nPriors + g_nPointPriors > 0) error += pe / (float)(nPriors + g_nPointPriors);   /* point priors: _error == 0 */
    free(JTJ); free(JTe);
    noobs_free(&c);
    return error;
}

float orc_optimize_poses_noobs(const orc_model *const *models, float *const *T_mcs, float *const *joints,
                               const orc_noobs_options *o, const float *const *clouds, const int32_t *nClouds,
                               const int32_t *const *potentials, const orc_contact_prior *priors, int nPriors,
                               float *iterInfo, float *firstSystem) {
    noobs_ctx c;
    orc_contact_prior *mp = priors_copy(priors, nPriors);      /* updatePriorParams moves the contact points */
    priors = mp;
    noobs_init(&c, models, o->nModels, priors, nPriors);
    const int sysN = c.sys, D = c.D;
    float *JTJ = (float *)malloc(sizeof(float) * (size_t)sysN * sysN);
    float *JTe = (float *)malloc(sizeof(float) * (size_t)sysN);
    float *x = (float *)malloc(sizeof(float) * (size_t)sysN);
    for (int it = 0; it < o->numIterations; ++it) {
        memset(JTJ, 0, sizeof(float) * (size_t)sysN * sysN);
        memset(JTe, 0, sizeof(float) * (size_t)sysN);
        noobs_set_poses(&c, models, T_mcs, joints);
        float eInt = 0.0f, ePri = 0.0f;
        int nActive = 0;
        for (int m = 0; m < c.n; ++m) {
            const int dims = c.dims[m], off = c.offs[m];
            float *eJ = (float *)calloc((size_t)dims, sizeof(float));
            float *J = (float *)calloc((size_t)dims * dims, sizeof(float));
            float *sys = (float *)malloc(sizeof(float) * (size_t)sys_len(6 + models[m]->n_joints));
            eInt += noobs_intersection_terms(&c, models, T_mcs, m, o, clouds, nClouds, potentials, eJ, J, sys);
            /* symmetrise, += damping, scatter the upper triangle (optimizer.cpp:881-905) */
            if (g_damping && g_damping[m])
                for (int i = 0; i < dims; ++i) for (int j = i; j < dims; ++j) J[i * dims + j] += g_damping[m][i * dims + j];
            for (int i = 0; i < dims; ++i) for (int j = i; j < dims; ++j)
                if (J[i * dims + j] != 0) JTJ[(off + i) * sysN + off + j] = J[i * dims + j];
            for (int i = 0; i < dims; ++i) JTe[off + i] = eJ[i];
            free(eJ); free(J); free(sys);
        }
        for (int p = 0; p < nPriors; ++p) { /* optimizer.cpp:913-916 */
            int32_t ds = -1;
            const float e = orc_contact_prior_contribution(&priors[p], models, c.cT_mc, c.cT_cm, c.cT_fm, c.cT_mf, c.offs, c.dims,
                                                           c.priorOff[p], JTJ, sysN, JTe, &ds);
            ePri += e;
            (void)ds;
        }
        for (int p = 0; p < g_nPointPriors; ++p)
            orc_point_prior_contribution(&g_pointPriors[p], models, c.cT_mc, c.cT_cm, c.cT_fm, c.cT_mf, c.offs, c.dims, JTJ, sysN, JTe);
        if (it == 0 && firstSystem) {
            for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) firstSystem[i * D + j] = JTJ[i * sysN + j];
            for (int i = 0; i < D; ++i) firstSystem[D * D + i] = JTe[i];
        }
        for (int m = 0; m < c.n; ++m) /* optimizer.cpp:919-929 */
            for (int i = 0; i < c.dims[m]; ++i) {
                float *d = &JTJ[(c.offs[m] + i) * sysN + c.offs[m] + i];
                *d += o->regularization[m];
                *d += o->regularizationScaled[m] * (*d);
            }
        orc_solve_upper(JTJ, JTe, sysN, x);
        for (int i = 0; i < sysN; ++i) x[i] = -x[i];
        for (int m = 0; m < c.n; ++m) { /* optimizer.cpp:933-954 */
            float dT[12];
            orc_se3_exp(x + c.offs[m], dT);
            orc_se3_mul(dT, T_mcs[m], T_mcs[m]);
            if (!noobs_red_type(m)) {
                for (int i = 0; i < models[m]->n_joints; ++i) {
                    const float v = joints[m][i] + x[c.offs[m] + 6 + i];
                    joints[m][i] = fminf(fmaxf(models[m]->joint_limits[2 * i], v), models[m]->joint_limits[2 * i + 1]);
                }
            } else {        /* reduced articulation, reduced limits (optimizer.cpp:948-950) */
                for (int i = 0; i < g_noobsRed[m].redArt; ++i)
                    joints[m][i] = fminf(fmaxf(g_noobsRed[m].mins[i], joints[m][i] + x[c.offs[m] + 6 + i]), g_noobsRed[m].maxs[i]);
            }
        }
        for (int p = 0; p < nPriors; ++p) /* optimizer.cpp:956-958 */
            orc_contact_prior_update_params(&mp[p], x + c.priorOff[p], models);
        if (iterInfo) {
            float *r = iterInfo + 8 * it;
            r[0] = eInt; r[1] = ePri; r[2] = (float)nActive; r[3] = r[4] = r[5] = r[6] = r[7] = 0;
        }
    }
    free(JTJ); free(JTe); free(x);
    noobs_free(&c);
    contacts_out(mp, nPriors);
    const float err = orc_get_error_noobs(models, T_mcs, joints, o, clouds, nClouds, potentials, mp, nPriors);
    free(mp);
    return err;
}

/* ------------------------------------------------------------------------------------------------
 * Optimizer::optimizePoses (optimizer.cpp:524-781)
 * ---------------------------------------------------------------------------------------------- */
static void project_reduced(const orc_model *m, const orc_reduction *r, const float *art, float *full) {
    const int nf = m->n_joints;
    if (!r || r->type == 0) { for (int f = 0; f < nf; ++f) full[f] = art[f]; return; }
    if (r->type == 2) { for (int f = 0; f < nf; ++f) full[f] = art[r->mapping[f]]; return; }   /* pose.cpp:91-97 */
    for (int f = 0; f < nf; ++f) {                                                              /* pose.cpp:66-73 */
        full[f] = r->b[f];
        for (int k = 0; k < r->redArt; ++k) full[f] += art[k] * r->A[k + f * r->redArt];
    }
}

void orc_optimize_poses(const orc_model *const *models, float *const *T_mcs, float *const *art, const orc_reduction *red,
                        const orc_multi_options *o, const float *vert, const float *norm, int width, int height,
                        const float *const *clouds, const int32_t *nClouds, const int32_t *const *potentials,
                        const orc_contact_prior *priors, int nPriors, float *const *obsSdf, float *iterInfo,
                        float *firstSystem) {
    const int n = o->nModels;
    int offs[ORC_MAX_MODELS], rdims[ORC_MAX_MODELS], D = 0;
    float *T_mf[ORC_MAX_MODELS], *T_fm[ORC_MAX_MODELS], *T_cm[ORC_MAX_MODELS], *full[ORC_MAX_MODELS], *T_obs[ORC_MAX_MODELS];
    orc_da_point *pts[ORC_MAX_MODELS];
    orc_grid obs[ORC_MAX_MODELS];
    const float *cT_mc[ORC_MAX_MODELS], *cT_cm[ORC_MAX_MODELS], *cT_fm[ORC_MAX_MODELS], *cT_mf[ORC_MAX_MODELS];
    int maxDims = 0;
    for (int m = 0; m < n; ++m) {
        const int redType = red ? red[m].type : 0;
        rdims[m] = 6 + (redType ? red[m].redArt : models[m]->n_joints);
        offs[m] = D; D += rdims[m];
        if (6 + models[m]->n_joints > maxDims) maxDims = 6 + models[m]->n_joints;
        T_mf[m] = (float *)malloc(sizeof(float) * 12 * (size_t)models[m]->n_frames);
        T_fm[m] = (float *)malloc(sizeof(float) * 12 * (size_t)models[m]->n_frames);
        T_cm[m] = (float *)malloc(sizeof(float) * 12);
        T_obs[m] = (float *)malloc(sizeof(float) * 12);
        full[m] = (float *)calloc((size_t)models[m]->n_joints + 1, sizeof(float));
        pts[m] = (orc_da_point *)malloc(sizeof(orc_da_point) * (size_t)width * height);
        obs[m].dim[0] = models[m]->obs_sdf_dim[0]; obs[m].dim[1] = models[m]->obs_sdf_dim[1]; obs[m].dim[2] = models[m]->obs_sdf_dim[2];
        obs[m].offset[0] = models[m]->obs_sdf_offset[0]; obs[m].offset[1] = models[m]->obs_sdf_offset[1]; obs[m].offset[2] = models[m]->obs_sdf_offset[2];
        obs[m].resolution = models[m]->obs_sdf_resolution;
        obs[m].data = obsSdf ? obsSdf[m] : NULL;
    }
    orc_contact_prior *mp = priors_copy(priors, nPriors);      /* updatePriorParams moves the contact points (:757-759) */
    priors = mp;
    int *priorOff = (int *)malloc(sizeof(int) * (size_t)(nPriors > 0 ? nPriors : 1));
    const int sysN = prior_param_offsets(priors, nPriors, D, priorOff);
    float *JTJ = (float *)malloc(sizeof(float) * (size_t)sysN * sysN);
    float *JTe = (float *)malloc(sizeof(float) * (size_t)sysN);
    float *x = (float *)malloc(sizeof(float) * (size_t)sysN);
    float *sys = (float *)malloc(sizeof(float) * (size_t)sys_len(maxDims));
    float *pred = (float *)calloc((size_t)o->predWidth * o->predHeight * 4, sizeof(float));
    float *T_mc_flat = (float *)malloc(sizeof(float) * 12 * (size_t)n);

    if (o->lambdaModToObs > 0) { /* generateObsSdf per model with its start pose, optimizer.cpp:574-580 */
        for (int m = 0; m < n; ++m) {
            const size_t nv = (size_t)obs[m].dim[0] * obs[m].dim[1] * obs[m].dim[2];
            float *tmp = (float *)malloc(sizeof(float) * nv);
            const float *T_start = g_obsStart ? g_obsStart + 12 * m : T_mcs[m];
            orc_se3_invert(T_start, T_cm[m]);
            orc_splat_obs_sdf(vert, width, height, T_cm[m], &obs[m], o->focalLength, obsSdf[m]);
            orc_distance_transform_3d(obsSdf[m], tmp, obs[m].dim[0], obs[m].dim[1], obs[m].dim[2], 1);
            memcpy(obsSdf[m], tmp, sizeof(float) * nv);
            free(tmp);
            memcpy(T_obs[m], T_start, sizeof(float) * 12);
        }
    }
    for (int it = 0; it < o->numIterations; ++it) {
        memset(JTJ, 0, sizeof(float) * (size_t)sysN * sysN);
        memset(JTe, 0, sizeof(float) * (size_t)sysN);
        for (int m = 0; m < n; ++m) { /* projectReducedToFull; setPose */
            project_reduced(models[m], red ? &red[m] : NULL, art[m], full[m]);
            orc_fk_set_pose(models[m], full[m], T_mf[m], T_fm[m]);
            orc_se3_invert(T_mcs[m], T_cm[m]);
            memcpy(T_mc_flat + 12 * m, T_mcs[m], sizeof(float) * 12);
            cT_mc[m] = T_mcs[m]; cT_cm[m] = T_cm[m]; cT_fm[m] = T_fm[m]; cT_mf[m] = T_mf[m];
        }
        if (o->lambdaModToObs > 0) { /* raytracePrediction (one pass per model) + cullUnobservable, :610-618 */
            for (int m = 0; m < n; ++m)
                orc_raycast_prediction(o->predFl[0], o->predFl[1], (float)(o->predWidth / 2), (float)(o->predHeight / 2), o->predWidth,
                                       o->predHeight, m, models[m], T_mcs[m], T_fm[m], T_mf[m], 0.0f, pred);
            orc_cull_unobservable(pred, o->predWidth, o->predHeight, vert, width, height);
        }
        int32_t counts[ORC_MAX_MODELS];
        float pn[ORC_MAX_MODELS * 3];
        for (int m = 0; m < n; ++m) { pn[3 * m] = o->planeNormal[m][0]; pn[3 * m + 1] = o->planeNormal[m][1]; pn[3 * m + 2] = o->planeNormal[m][2]; }
        orc_assoc_multi(vert, norm, width, height, n, models, T_mc_flat, (const float *const *)T_fm, o->distThreshold, o->planeOffset, pn,
                        pts, counts, NULL, NULL);
        for (int m = 0; m < n; ++m) {
            const int dims = rdims[m], off = offs[m], fullDims = 6 + models[m]->n_joints;
            const int redType = red ? red[m].type : 0;
            const float *A = redType == 1 ? red[m].A : NULL;
            const int32_t *map = redType == 2 ? red[m].mapping : NULL;
            float *eJ = (float *)calloc((size_t)dims, sizeof(float));
            float *J = (float *)calloc((size_t)dims * dims, sizeof(float));
            float eO2M = 0, eM2O = 0, eInt = 0;
            int nPred = 0;
            if (o->lambdaObsToMod > 0) {
                orc_ne_obs2mod(vert, models[m], T_mcs[m], T_fm[m], T_mf[m], pts[m], counts[m], o->huberDelta, redType, dims, A, map, sys);
                eO2M = orc_unpack(eJ, J, dims, sys, o->lambdaObsToMod, dims);
            }
            if (o->lambdaModToObs > 0) { /* reduced wrappers use the model's CURRENT T_mc, the full one T_obsSdf_c (modToObs.cu:851-988) */
                nPred = orc_ne_mod2obs(pred, o->predWidth, o->predHeight, m, models[m], redType ? T_mcs[m] : T_obs[m], T_fm[m], T_mf[m], &obs[m],
                                       -1.0f, redType, dims, A, map, sys);
                eM2O = orc_unpack(eJ, J, dims, sys, o->lambdaModToObs, dims);
            }
            if (o->lambdaIntersection[m + m * n] > 0 && dims > 6) {
                const int ad = dims - 6;
                orc_ne_self_intersection(clouds[m], nClouds[m], models[m], T_fm[m], T_mf[m], potentials[m], redType, dims, A, map, sys);
                float *eJt = (float *)calloc((size_t)ad, sizeof(float));
                float *Jt = (float *)calloc((size_t)ad * ad, sizeof(float));
                eInt = orc_unpack(eJt, Jt, ad, sys, o->lambdaIntersection[m + m * n], ad);
                for (int i = 0; i < ad; ++i) {
                    eJ[6 + i] += eJt[i];
                    for (int j = 0; j < ad; ++j) J[(6 + i) * dims + 6 + j] += Jt[i * ad + j];
                }
                free(eJt); free(Jt);
            }
            for (int d = 0; d < n; ++d) {
                if (d == m || !(o->lambdaIntersection[m + d * n] > 0)) continue;
                float T_ds[12], T_sd[12];
                orc_se3_mul(T_mcs[d], T_cm[m], T_ds);
                orc_se3_invert(T_ds, T_sd);
                orc_ne_intersection(clouds[m], nClouds[m], T_ds, T_sd, models[m], T_fm[m], T_mf[m], models[d], T_fm[d], T_mf[d], redType, dims,
                                    A, map, sys);
                eInt = orc_unpack(eJ, J, dims, sys, o->lambdaIntersection[m + d * n], dims);
            }
            (void)fullDims;
            /* symmetrise; += reg*I; A_ii += lm*A_ii; += damping; scatter the upper triangle (:688-714) */
            for (int i = 0; i < dims; ++i) for (int j = 0; j < i; ++j) J[i * dims + j] = J[j * dims + i];
            for (int i = 0; i < dims; ++i) J[i * dims + i] += o->regularization[m] * 1.0f;
            for (int i = 0; i < dims; ++i) J[i * dims + i] += o->regularizationScaled[m] * J[i * dims + i];
            if (g_damping && g_damping[m])
                for (int i = 0; i < dims * dims; ++i) J[i] += g_damping[m][i];
            for (int i = 0; i < dims; ++i) for (int j = i; j < dims; ++j)
                if (J[i * dims + j] != 0) JTJ[(off + i) * sysN + off + j] = J[i * dims + j];
            for (int i = 0; i < dims; ++i) JTe[off + i] = eJ[i];
            if (iterInfo) {
                float *r = iterInfo + ((size_t)it * n + m) * 8;
                r[0] = eO2M; r[1] = eM2O; r[2] = eInt; r[3] = (float)counts[m]; r[4] = (float)nPred; r[5] = r[6] = r[7] = 0;
            }
            free(eJ); free(J);
        }
        for (int p = 0; p < nPriors; ++p) /* :727-730 */
            orc_contact_prior_contribution(&priors[p], models, cT_mc, cT_cm, cT_fm, cT_mf, offs, rdims, priorOff[p], JTJ, sysN, JTe, NULL);
        if (it == 0 && firstSystem) {
            for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) firstSystem[i * D + j] = JTJ[i * sysN + j];
            for (int i = 0; i < D; ++i) firstSystem[D * D + i] = JTe[i];
        }
        orc_solve_upper(JTJ, JTe, sysN, x);
        for (int i = 0; i < sysN; ++i) x[i] = -x[i];
        for (int m = 0; m < n; ++m) { /* :734-755 */
            float dT[12];
            orc_se3_exp(x + offs[m], dT);
            orc_se3_mul(dT, T_mcs[m], T_mcs[m]);
            const int redType = red ? red[m].type : 0;
            const int na = rdims[m] - 6;
            for (int i = 0; i < na; ++i) {
                const float lo = redType ? red[m].mins[i] : models[m]->joint_limits[2 * i];
                const float hi = redType ? red[m].maxs[i] : models[m]->joint_limits[2 * i + 1];
                art[m][i] = fminf(fmaxf(lo, art[m][i] + x[offs[m] + 6 + i]), hi);
            }
        }
        for (int p = 0; p < nPriors; ++p) orc_contact_prior_update_params(&mp[p], x + priorOff[p], models);     /* :757-759 */
    }
    contacts_out(mp, nPriors);
    free(mp); free(priorOff);
    for (int m = 0; m < n; ++m) { free(T_mf[m]); free(T_fm[m]); free(T_cm[m]); free(T_obs[m]); free(full[m]); free(pts[m]); }
    free(JTJ); free(JTe); free(x); free(sys); free(pred); free(T_mc_flat);
}
