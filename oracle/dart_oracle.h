/*
 * dart_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic of contactgrasp/dart's Gauss-Newton hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * nothing under dart_amd/ links, imports or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * The reference itself is unbuildable in this image (every hot-path source needs CUDA headers,
 * helper_math.h from the CUDA samples, Eigen, tinyxml or GL), so this restatement is pinned by the
 * reference's own known-answer tests (test/test_obs_to_mod_kernels.cpp, test/test_voxelize.cpp,
 * test/test_model_jacobian.{cpp,cu}, test/test_pose_reduction.cpp), re-stated in tests/.
 * Normal-equation values, intersection, contact priors and the solve are covered by NO reference
 * test: for those the parity is "unpinned" (restatement only), as DESIGN.md says.
 *
 * Conventions: SE3 = float[12] row-major [R|t] (SE3.h:17-23); grids hold voxel-unit values, x fastest
 * (grid_3d.h:82-84). Per-point arithmetic is fp32 in the reference's operation order (compile with
 * -ffp-contract=off); sums over points are accumulated in double and emitted as float.
 */
#ifndef DART_ORACLE_H
#define DART_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t dim[3];
    float offset[3];
    float resolution;
    const float *data;
} orc_grid;

/* same field order as dartk_model_desc (include/dartk.h) so tests can fill both from one set of arrays */
typedef struct {
    int32_t n_frames, n_joints, n_sdfs, model_id;
    const int32_t *frame_parent;   /* [n_frames] */
    const int32_t *joint_type;     /* [n_joints] 0 rotational, 1 prismatic (dart_types.h:6-9) */
    const float *joint_axis;       /* [n_joints*3] */
    const float *T_pf;             /* [n_joints*12] joint-axis-to-parent */
    const float *dh;               /* [n_joints*4] a, alpha, d, theta */
    const float *joint_limits;     /* [n_joints*2] */
    const int32_t *sdf_frame;      /* [n_sdfs] */
    const int32_t *dependencies;   /* [n_frames*n_joints] */
    const int32_t *sdf_dim;        /* [n_sdfs*3] */
    const float *sdf_offset;       /* [n_sdfs*3] */
    const float *sdf_resolution;   /* [n_sdfs] */
    const float *const *sdf_data;  /* [n_sdfs] */
    int32_t obs_sdf_dim[3];
    float obs_sdf_offset[3];
    float obs_sdf_resolution;
} orc_model;

typedef struct { int32_t index; int32_t dataAssociation; float error; } orc_da_point; /* optimization.h:85-89 */

/* ---- lookup statistics (measurement aid: number of trilinear lookups per caller category) ---- */
enum { ORC_STAT_OTHER = 0, ORC_STAT_ASSOC = 1, ORC_STAT_ASSOC_GRAD = 2, ORC_STAT_NE_OBS2MOD = 3, ORC_STAT_RAYCAST = 4,
       ORC_STAT_NE_MOD2OBS = 5, ORC_STAT_INTERSECTION = 6, ORC_STAT_PRIOR = 7, ORC_STAT_CATEGORIES = 8 };
/* worker threads of the per-pixel loops (OpenMP); 1 = the serial restatement (default, what every parity test uses) */
void orc_set_threads(int n);
int orc_get_threads(void);
void orc_stats_reset(void);
void orc_stats_enable(int on);
void orc_stats_get(long long *out /*[ORC_STAT_CATEGORIES]*/);

/* ---- SE3 (SE3.h) ---- */
void orc_se3_mul(const float *A, const float *B, float *out);          /* SE3.h:396-409 */
void orc_se3_invert(const float *A, float *out);                       /* SE3.h:447-451 */
void orc_se3_exp(const float *t6, float *out);                         /* SE3.h:147-180 */
void orc_se3_log(const float *A, float *t6);                           /* SE3.h:250-299 */
void orc_se3_from_euler(float phi, float theta, float psi, float *out);/* SE3.h:134-144 */
void orc_se3_from_dh_graspit(float theta, float d, float a, float alpha, float yrot, float *out); /* SE3.h:103-106 */

/* ---- Grid3D (grid_3d.h) ---- */
int   orc_grid_in_bounds_interp(const orc_grid *g, const float *pg);           /* :63-67 */
int   orc_grid_in_bounds_gradient_interp(const orc_grid *g, const float *pg);  /* :76-80 */
void  orc_grid_coords(const orc_grid *g, const float *pw, float *pg);          /* :49-51 */
float orc_grid_value_interp(const orc_grid *g, const float *pg);               /* :86-112 */
void  orc_grid_gradient_interp(const orc_grid *g, const float *pg, float *grad);/* :314-330 */

/* ---- forward kinematics ---- */
/* MirroredModel::setPose, mirrored_model.cpp:185-225 (GraspIt DH form) */
void orc_fk_set_pose(const orc_model *m, const float *joints, float *T_mf, float *T_fm);
/* MirroredModel::setArticulation / HostOnlyModel::setPose, mirrored_model.cpp:154-183 (axis-angle form) */
void orc_fk_set_articulation(const orc_model *m, const float *joints, float *T_mf, float *T_fm);

/* ---- Jacobians ---- */
/* kernel_common.h:12-45 (articulation_only=0) / :47-71 (=1, J has n_joints entries) */
void orc_error_jacobian(const orc_model *m, const float *T_fm, const float *T_mf, const float *point_m,
                        int frame, const float *grad_m, int articulation_only, float *J);
/* Model::getModelJacobianOfModelPoint, model.cpp:419-493; J3D is [dims*3] */
void orc_model_jacobian_3d(const orc_model *m, const float *T_fm, const float *T_mf, const float *point_m,
                           int frame, float *J3D);

/* ---- obs -> mod ---- */
/* gpu_errorAndDataAssociationObsToMod, obsToMod.cu:12-114. pts ordered by pixel index. dbg* may be NULL. */
int orc_assoc_single(const float *vert, const float *norm, int width, int height, const orc_model *m,
                     const float *T_mc, const float *T_fm, float distThreshold, float normThreshold,
                     float planeOffset, const float *planeNormal, orc_da_point *pts,
                     int32_t *dbgDA, float *dbgErr, float *dbgNorm);
/* gpu_errorAndDataAssociationObsToModMultiModel, obsToMod.cu:116-244. pts[m] each sized width*height;
 * counts[m] written. dbgDA = (model<<16)|grid. */
void orc_assoc_multi(const float *vert, const float *norm, int width, int height, int nModels,
                     const orc_model *const *models, const float *T_mcs, const float *const *T_fms,
                     const float *distThresholds, const float *planeOffsets, const float *planeNormals,
                     orc_da_point *const *pts, int32_t *counts, int32_t *dbgDA, float *dbgErr);
/* gpu_normEqnsObsToMod{,Reduced,ParamMap}, obsToMod.cu:246-501; host wrappers :621-817 (n<=10 => zeros).
 * reduction: 0 none, 1 linear (A = dtheta_dalpha [ (redDims-6) + (fullDims-6)... ] layout kernel_common.h:84),
 * 2 param map (mapping[fullDims-6]). result has redDims + JTJSize(redDims) + 1 floats. */
void orc_ne_obs2mod(const float *vert, const orc_model *m, const float *T_mc, const float *T_fm,
                    const float *T_mf, const orc_da_point *pts, int nPoints, float huberDelta,
                    int reduction, int redDims, const float *dtheta_dalpha, const int32_t *mapping,
                    float *result);

/* ---- mod -> obs ---- */
/* gpu_raycastPrediction, raycast.cu:121-215 (one call per model, modelNum 0 clears) */
void orc_raycast_prediction(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                            const orc_model *m, const float *T_mc, const float *T_fm, const float *T_mf,
                            float levelSet, float *prediction);
/* gpu_cullUnobservable, modToObs.cu:682-709 */
void orc_cull_unobservable(float *pred, int pw, int ph, const float *obsVert, int ow, int oh);
/* gpu_depthToVertices<float>, img_proc/organized_point_cloud.cu:11-40 (out-of-range depth: only .w is cleared) */
void orc_depth_to_vertices(const float *depth, float *vert, int width, int height, float ppx, float ppy, float flx,
                           float fly, float rangeMin, float rangeMax);
/* gpu_verticesToNormals, img_proc/organized_point_cloud.cu:265-305 */
void orc_vertices_to_normals(const float *vert, float *norm, int width, int height);
/* gpu_splatObsSdf, modToObs.cu:403-441 (T_cm = model-to-camera) */
void orc_splat_obs_sdf(const float *vert, int width, int height, const float *T_cm, const orc_grid *obsSdf,
                       float focalLength, float *data);
/* distanceTransform3D<float,takeSqrt>, distance_transforms.cu:434-458 (+ :38-126 for the 1-D pass).
 * 'in' is clobbered exactly as in the reference (z pass writes back into it). */
void orc_distance_transform_3d(float *in, float *out, int width, int height, int depth, int takeSqrt);
/* signedDistanceTransform3D<float,takeSqrt>, distance_transforms.cu:540-562 */
void orc_signed_distance_transform_3d(float *in, float *out, int width, int height, int depth, int takeSqrt);
/* gpu_normEqnsModToObs{,Truncated,Reduced,ParamMap}, modToObs.cu:14-401. truncationDist<0 => untruncated.
 * T_gc = camera -> obs-SDF frame. Returns numPredictions. */
int orc_ne_mod2obs(const float *pred, int width, int height, int modelNum, const orc_model *m,
                   const float *T_gc, const float *T_fm, const float *T_mf, const orc_grid *obsSdf,
                   float truncationDist, int reduction, int redDims, const float *dtheta_dalpha,
                   const int32_t *mapping, float *result);
/* gpu_errorModToObs, modToObs.cu:649-680 */
float orc_error_mod2obs(const float *pred, int width, int height, const orc_grid *obsSdf);

/* ---- intersection ---- */
/* gpu_normEqnsSelfIntersection{,Reduced,ParamMap}, intersection.cu:66-300; result (redDims-6)-sized system */
int orc_ne_self_intersection(const float *sites, int nSites, const orc_model *m, const float *T_fm,
                             const float *T_mf, const int32_t *potentialIntersection, int reduction,
                             int redDims, const float *dtheta_dalpha, const int32_t *mapping, float *result);
/* gpu_normEquationsIntersection{,Reduced,ParamMap}, intersection.cu:367-597 */
int orc_ne_intersection(const float *sites, int nSites, const float *T_ds, const float *T_sd,
                        const orc_model *src, const float *T_fm_src, const float *T_mf_src,
                        const orc_model *dst, const float *T_fm_dst, const float *T_mf_dst, int reduction,
                        int redDims, const float *dtheta_dalpha, const int32_t *mapping, float *result);
/* gpu_selfIntersectionCount :11-64 / gpu_intersectionCount :315-365 */
/* the kernels' optional debug outputs (modToObs.cu:47-82,145-186; intersection.cu:131-133,303-313,433-435) */
void orc_mod2obs_debug_maps(const float *pred, int width, int height, int modelNum, const float *T_gc, const orc_grid *obsSdf,
                            float truncationDist, int32_t *dbgDA, float *dbgErr, float *dbgNorm);
void orc_init_debug_intersection_error(float *debugError, int nSites);
void orc_debug_self_intersection_error(const float *sites, int nSites, const orc_model *m, const float *T_fm, const float *T_mf,
                                       const int32_t *pot, float *debugError);
void orc_debug_intersection_error(const float *sites, int nSites, const float *T_ds, const orc_model *src, const float *T_mf_src,
                                  const orc_model *dst, const float *T_fm_dst, float *debugError);
int orc_count_self_intersections(const float *sites, int nSites, const orc_model *m, const float *T_fm,
                                 const float *T_mf, const int32_t *potentialIntersection);
int orc_count_intersections(const float *sites, int nSites, const float *T_ds, const orc_model *src,
                            const float *T_mf_src, const orc_model *dst, const float *T_fm_dst);

/* ---- host-side assembly / solve (optimizer.cpp) ---- */
/* OptimizerNoObs::unpack, optimizer.cpp:50-70: eJ += mult*sys, JTJ upper (i<=j) += mult*..., returns mult*e.
 * JTJ is dense row-major [dims*dims]. */
float orc_unpack(float *eJ, float *JTJ, int ld, const float *sys, float multiplier, int dims);
/* Eigen MatrixXf::ldlt().solve (optimizer.cpp:501): robust Cholesky with diagonal pivoting, restated from
 * Eigen's documented algorithm (parity unpinned: Eigen is not vendored). A symmetric [n*n], x = A^-1 b. */
void orc_solve_ldlt(const float *A, const float *b, int n, float *x);
/* SparseMatrix::triangularView<Upper>().solve (optimizer.cpp:732,931): back substitution on the upper
 * triangle, column-oriented, skipping columns whose current rhs entry is exactly 0. */
void orc_solve_upper(const float *A, const float *b, int n, float *x);

/* ---- contact prior (contact_prior.cpp:8-205) ---- */
typedef struct {
    int32_t srcModel, dstModel, srcSdf;
    float weight;
    float contactPoint[3];   /* source SDF frame */
    float normal[3];         /* _normal_sf */
    int32_t invert;          /* repulsive */
    float contactThreshold;
    int32_t updateSrc, updateDst, updateContact;
} orc_contact_prior;
/* Evaluates one prior. Adds into dense upper-triangular JTJ [sys*sys] (row-major) and JTe [sys] using
 * modelOffsets/srcDims/dstDims exactly as computeContribution does for unreduced poses; returns _error.
 * dstSdfOut receives the selected destination SDF (-1 if none). */
float orc_contact_prior_contribution(const orc_contact_prior *p, const orc_model *const *models,
                                     const float *const *T_mcs, const float *const *T_cms,
                                     const float *const *T_fms, const float *const *T_mfs,
                                     const int32_t *modelOffsets, const int32_t *modelDims,
                                     int priorParamOffset, float *JTJ, int sys, float *JTe,
                                     int32_t *dstSdfOut);

/* projectToSdfSurface (sdf.cpp:6-23) on grid coordinates, and ContactPrior::updatePriorParams (contact_prior.cpp:207-221) */
void orc_project_to_sdf_surface(const orc_grid *sdf, float *pointGrid, float threshold, int maxIters);
void orc_contact_prior_update_params(orc_contact_prior *p, const float *update, const orc_model *const *models);

/* ---- keypoint priors (point_3d_3d_prior.cpp:6-69, point_2d_3d_prior.cpp:6-83) ---- */
typedef struct {
    int32_t kind;               /* 0 Point3D3DPrior, 1 Point2D3DPrior */
    int32_t srcModel, srcFrame;
    float weight;
    float point_f[3];           /* _point_f */
    float point_c[3];           /* _point_c (kind 1: pixel x, y) */
    float focalLength[2], principalPoint[2];
} orc_point_prior;
/* Evaluates one prior: adds _weight * subJ^T subJ (upper triangle, zeros skipped) and _weight * subJ^T subE into the
 * source model's block. These priors never set _error (it stays 0). */
void orc_point_prior_contribution(const orc_point_prior *p, const orc_model *const *models, const float *const *T_mcs,
                                  const float *const *T_cms, const float *const *T_fms, const float *const *T_mfs,
                                  const int32_t *modelOffsets, const int32_t *modelDims, float *JTJ, int sys, float *JTe);

#ifdef __cplusplus
}
#endif
#endif
