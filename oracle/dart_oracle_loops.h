/* dart_oracle_loops.h -- CPU ORACLE (test infrastructure): optimizer loops. See dart_oracle.h. */
#ifndef DART_ORACLE_LOOPS_H
#define DART_ORACLE_LOOPS_H
#include "dart_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

/* same field order as dartk_opt_options (include/dartk.h); values of OptimizationOptions (optimization.h:9-72) */
typedef struct {
    int32_t numIterations;
    float distThreshold, normThreshold, planeOffset;
    float planeNormal[3];
    float regularization, regularizationScaled;
    float lambdaObsToMod, lambdaModToObs, lambdaIntersection;
    float huberDelta, focalLength;
    int32_t predWidth, predHeight;
    float predFl[2];
} orc_opt_options;

/* Optimizer::optimizePose (optimizer.cpp:423-522). T_mc/joints are updated in place. obsSdfData: scratch of
 * the model's obs-SDF size. iterInfo (optional): 8 floats per iteration {errObsToMod, errModToObs, errIntersection,
 * nAssociated, nPredicted, 0,0,0}. firstSystem (optional): the first iteration's packed systems
 * [obs->mod | mod->obs | self-intersection]. */
/* test aid: the next optimize_pose(s) calls build their observation SDFs at these poses ([nModels][12]); NULL = own start poses */
void orc_set_obs_sdf_start(const float *T_mcs);
void orc_optimize_pose(const orc_model *m, float *T_mc, float *joints, const float *vert, const float *norm,
                       int width, int height, const orc_opt_options *o, const float *cloud, int nCloud,
                       const int32_t *potential, float *obsSdfData, float *iterInfo, float *firstSystem);

/* ---- Optimizer::optimizePoses (multi-model tracking with observations) ---- */
typedef struct {
    int32_t type;            /* 0 NullReduction, 1 LinearPoseReduction, 2 ParamMapPoseReduction (pose_reduction.h) */
    int32_t redArt;          /* reduced articulated dimensions */
    const float *A;          /* [redArt * fullArt], A[r + f*redArt] (pose.cpp:66-73) = the device dtheta_dalpha */
    const float *b;          /* [fullArt] */
    const int32_t *mapping;  /* [fullArt] (type 2) */
    const float *mins, *maxs;/* [redArt] reduced limits */
} orc_reduction;

typedef struct {
    int32_t numIterations, nModels;
    float distThreshold[8], planeOffset[8], planeNormal[8][3];
    float regularization[8], regularizationScaled[8];
    float lambdaObsToMod, lambdaModToObs;
    float lambdaIntersection[64];
    float huberDelta, focalLength;
    int32_t predWidth, predHeight;
    float predFl[2];
} orc_multi_options;

/* Optimizer::optimizePoses (optimizer.cpp:524-781), zero damping matrices. T_mcs[m] and art[m] (the REDUCED articulation
 * of model m; the full one for NullReduction) are updated in place. obsSdf[m]: scratch of model m's obs-SDF size.
 * iterInfo (optional): [iteration][model][8] = {errObsToMod, errModToObs, errIntersection (last term), nAssociated,
 * nPredicted, 0,0,0}. firstSystem (optional): dense upper JTJ [D*D] + JTe [D] of iteration 0 right before the solve. */
void orc_optimize_poses(const orc_model *const *models, float *const *T_mcs, float *const *art, const orc_reduction *red,
                        const orc_multi_options *o, const float *vert, const float *norm, int width, int height,
                        const float *const *clouds, const int32_t *nClouds, const int32_t *const *potentials,
                        const orc_contact_prior *priors, int nPriors, float *const *obsSdf, float *iterInfo,
                        float *firstSystem);

/* ---- OptimizerNoObs (ContactGrasp refinement) ---- */
#define ORC_MAX_MODELS 8
/* pose reductions for the following NoObs calls ([nModels] or NULL); reduced models pass their reduced articulation as joints[m] */
void orc_noobs_set_reductions(const orc_reduction *red);
typedef struct {
    int32_t numIterations;
    int32_t nModels;
    float regularization[ORC_MAX_MODELS];          /* opts.regularization[m] */
    float regularizationScaled[ORC_MAX_MODELS];    /* opts.regularizationScaled[m] (LM damping) */
    float lambdaIntersection[ORC_MAX_MODELS * ORC_MAX_MODELS];   /* [src + dst*nModels] (optimizer.cpp:866,873) */
} orc_noobs_options;

/* OptimizerNoObs::optimizePoses (optimizer.cpp:783-983) for NullReduction poses and zero damping matrices
 * (tracker_no_obs.cpp:214-215). T_mcs[m] (12 floats) and joints[m] are updated in place; the return value is
 * getError() (optimizer.cpp:985-1076) at the final poses. Prior-parameter rows exist only for priors with
 * updateContact set; for the others the reference's rows/columns are identically zero and its triangular solve
 * skips them (SURVEY A-1/A-7). iterInfo (optional): 8 floats per iteration {sum of intersection e (lambda-scaled),
 * sum of prior _error, nActivePriors, 0...}. firstSystem (optional): dense upper-triangular JTJ [D*D] followed by
 * JTe [D] of the first iteration, before regularisation, D = sum of model dims. */
float orc_optimize_poses_noobs(const orc_model *const *models, float *const *T_mcs, float *const *joints,
                               const orc_noobs_options *o, const float *const *clouds, const int32_t *nClouds,
                               const int32_t *const *potentials, const orc_contact_prior *priors, int nPriors,
                               float *iterInfo, float *firstSystem);
/* Keypoint priors for the two NoObs functions below: evaluated after the contact priors (the reference walks one mixed
 * list in insertion order; the order only fixes the order of the sums) and counted in getError's division by
 * priors.size(). The pointer is kept, not copied; n = 0 clears it. */
void orc_noobs_set_point_priors(const orc_point_prior *priors, int n);
/* OptimizerNoObs::getError (optimizer.cpp:985-1076) */
float orc_get_error_noobs(const orc_model *const *models, float *const *T_mcs, float *const *joints,
                          const orc_noobs_options *o, const float *const *clouds, const int32_t *nClouds,
                          const int32_t *const *potentials, const orc_contact_prior *priors, int nPriors);

/* orc_optimize_poses_noobs / orc_optimize_poses write the contact points of their priors after the last iteration here
 * ([nPriors][3]; priors with update_contact_point move, contact_prior.cpp:207-221). The pointer is kept; NULL clears it. */
void orc_set_contact_out(float *out);
/* damping matrices of the trackers (tracker.h:122, tracker_no_obs.h:73; optimizer.cpp:702-703, 895-896): [nModels] pointers to
 * [dims][dims] row-major matrices, NULL entries / NULL list = zero; stays in force for both loops until reset */
void orc_set_damping(const float *const *mats);

#ifdef __cplusplus
}
#endif
#endif
