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
void orc_optimize_pose(const orc_model *m, float *T_mc, float *joints, const float *vert, const float *norm,
                       int width, int height, const orc_opt_options *o, const float *cloud, int nCloud,
                       const int32_t *potential, float *obsSdfData, float *iterInfo, float *firstSystem);

#ifdef __cplusplus
}
#endif
#endif
