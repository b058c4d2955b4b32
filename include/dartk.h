/*
 * dartk.h -- C ABI of libdartk.so: the MI355X (gfx950) implementation of DART's Gauss-Newton hot path.
 *
 * This is the drop-in boundary: every entry point replaces one of the reference's `namespace dart` free
 * functions (or the MirroredModel device state they read); the comment above each cites the interface it
 * replaces as file:line relative to the reference tree. Plain pointers and sizes only; `d_` arguments are
 * device pointers, `h_` host pointers, `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 on success or a negative dartk_status; dartk_last_error() describes the failure.
 * Like the reference (optimizer.cpp:196-199, static scratch) a model handle is not re-entrant: calls that
 * touch one handle must be issued from one host thread / one stream at a time.
 *
 * Conventions (reference Appendix-C layouts): SE3 = float[12] row-major [R|t] (SE3.h:17-23); vertex /
 * normal / prediction maps = float4 per pixel (w>0 valid; prediction z==0 none, w = (model<<16)+sdf);
 * packed system = float[dims + dims(dims+1)/2 + 1] = [J^T e | lower-tri J^T J at i(i+1)/2+j | e]
 * (kernel_common.h:141-153, optimization.h:74).
 */
#ifndef DARTK_H
#define DARTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum dartk_status {
    DARTK_OK = 0,
    DARTK_ERR_INVALID = -1,   /* bad argument / shape mismatch (checked on the host before any launch) */
    DARTK_ERR_HIP = -2,       /* a HIP runtime call failed */
    DARTK_ERR_UNSUPPORTED = -3,
    DARTK_ERR_NO_DEVICE = -4
};

#define DARTK_MAX_JOINTS 64   /* dependency rows are one 64-bit mask per frame */
#define DARTK_MAX_MODELS 8

/* optimization.h:85-89 */
typedef struct { int32_t index; int32_t dataAssociation; float error; } dartk_da_point;

typedef struct dartk_model_s *dartk_model_t;

/* Everything MirroredModel's constructor copies to the device (mirrored_model.cpp:8-120). Host arrays. */
typedef struct {
    int32_t n_frames, n_joints, n_sdfs, model_id;
    const int32_t *frame_parent;   /* [n_frames], root = -1 */
    const int32_t *joint_type;     /* [n_joints] 0 rotational / 1 prismatic (dart_types.h:6-9) */
    const float *joint_axis;       /* [n_joints*3] */
    const float *T_pf;             /* [n_joints*12] joint-axis-to-parent transforms */
    const float *dh;               /* [n_joints*4] a, alpha, d, theta (host_only_model.cpp:120-125) */
    const float *joint_limits;     /* [n_joints*2] min,max */
    const int32_t *sdf_frame;      /* [n_sdfs] */
    const int32_t *dependencies;   /* [n_frames*n_joints] (host_only_model.cpp:97-103) */
    const int32_t *sdf_dim;        /* [n_sdfs*3] */
    const float *sdf_offset;       /* [n_sdfs*3] */
    const float *sdf_resolution;   /* [n_sdfs] */
    const float *const *sdf_data;  /* [n_sdfs] host pointers, voxel units, x fastest (grid_3d.h:82-84) */
    int32_t obs_sdf_dim[3];        /* observation SDF (mirrored_model.cpp:111-118); dims 0 = none */
    float obs_sdf_offset[3];
    float obs_sdf_resolution;
} dartk_model_desc;

/* ------------------------------------------------------------------ library / device */
int dartk_init(int device);                 /* hipSetDevice + workspace allocation; idempotent per device */
const char *dartk_last_error(void);
int dartk_version(void);

/* MirroredVector<T> device side (util/mirrored_memory.h:36-83) */
int dartk_buf_alloc(void **d_ptr, size_t bytes);
int dartk_buf_free(void *d_ptr);
int dartk_buf_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int dartk_buf_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int dartk_buf_zero(void *d_ptr, size_t bytes, void *stream);
int dartk_stream_sync(void *stream);

/* ------------------------------------------------------------------ model (src/model/mirrored_model.*) */
int dartk_model_create(const dartk_model_desc *desc, dartk_model_t *out);        /* mirrored_model.cpp:8-120 */
int dartk_model_destroy(dartk_model_t m);                                        /* :122-152 */
/* MirroredModel::setPose (mirrored_model.cpp:185-225): clamps joints, GraspIt-DH forward kinematics ON DEVICE,
 * fills T_mf/T_fm/T_mc/T_cm. h_T_mc = camera->model. */
int dartk_model_set_pose(dartk_model_t m, const float *h_T_mc, const float *h_joints, void *stream);
/* MirroredModel::setArticulation (mirrored_model.cpp:154-183): axis-angle FK variant; T_mc untouched. */
int dartk_model_set_articulation(dartk_model_t m, const float *h_joints, void *stream);
/* Upload host-computed transforms verbatim (what _T_mf/_T_fm->syncHostToDevice() does, :222-224). */
int dartk_model_set_transforms(dartk_model_t m, const float *h_T_mc, const float *h_T_mf, const float *h_T_fm,
                               void *stream);
int dartk_model_get_transforms(dartk_model_t m, float *h_T_mc, float *h_T_mf, float *h_T_fm, float *h_joints);
int dartk_model_num_sdfs(dartk_model_t m);
int dartk_model_dims(dartk_model_t m);
/* device pointer of the observation SDF data (MirroredModel::getDeviceObsSdfData) */
float *dartk_model_obs_sdf_data(dartk_model_t m);

/* MirroredModel's device getters (mirrored_model.h:65-87: getDeviceTransformsFrameToModel / ...ModelToFrame / getDeviceSdfFrames /
 * getDeviceSdfs / getDeviceFrameParents / getDeviceJointTypes / getDeviceJointAxes): the library's own device arrays, for
 * host code that collects them into per-model pointer tables (tracker.cpp:238-243) and hands those to the kernels. SE3 arrays
 * have the reference layout (12 floats per frame); the SDF table is the library's own header type (opaque to host code, as
 * Grid3D<float>* is in the reference: it only ever travels back into kernels). */
const float *dartk_model_device_T_mf(dartk_model_t m);
const float *dartk_model_device_T_fm(dartk_model_t m);
const float *dartk_model_device_T_mc(dartk_model_t m);
const int32_t *dartk_model_device_sdf_frames(dartk_model_t m);
const void *dartk_model_device_sdfs(dartk_model_t m);
const int32_t *dartk_model_device_frame_parents(dartk_model_t m);
const int32_t *dartk_model_device_joint_types(dartk_model_t m);
const float *dartk_model_device_joint_axes(dartk_model_t m);
/* The live model that owns a device array returned by one of the getters above (NULL if none): lets an entry point that
 * receives raw arrays recover the handle. */
dartk_model_t dartk_model_from_device_pointer(const void *d_ptr);

/* ------------------------------------------------------------------ obs -> mod (kernels/obsToMod.h) */
/* errorAndDataAssociation, obsToMod.h:75-86 / obsToMod.cu:504-555. Zeroes *d_last, appends associated points,
 * copies the count to *h_last (blocking, like the reference) when h_last != NULL. Debug maps may be NULL. */
int dartk_assoc_single(const float *d_vert, const float *d_norm, int width, int height, dartk_model_t m,
                       float distThreshold, float normThreshold, float planeOffset, const float *h_planeNormal3,
                       dartk_da_point *d_pts, int32_t *d_last, int32_t *h_last, int32_t *d_dbgDA,
                       float *d_dbgErr, float *d_dbgNorm, void *stream);
/* errorAndDataAssociationMultiModel, obsToMod.h:88-107 / obsToMod.cu:557-619. The per-model device pointer
 * tables of the reference (T_fms, sdfFrames, sdfs, nSdfs) are the model handles; thresholds/planes are host
 * arrays [nModels]; d_pts_table = device array of nModels device pointers. */
int dartk_assoc_multi(const float *d_vert, const float *d_norm, int width, int height, int nModels,
                      const dartk_model_t *models, const float *h_distThresholds, const float *h_planeOffsets,
                      const float *h_planeNormals3, dartk_da_point *const *h_pts_table, int32_t *d_last,
                      int32_t *d_dbgDA, float *d_dbgErr, void *stream);
/* errorAndDataAssociationMultiModel with the reference's OWN argument list (obsToMod.h:88-107; call-site optimizer.cpp:620-630):
 * every pointer is a device pointer, the per-model tables (T_fms, sdfFrames, sdfs, pts) are device arrays of device pointers
 * built from the getters above. The handles are recovered from T_fms[m]; T_mcs[m] (optimizer.cpp:603-608), when not null, is the
 * camera->model transform THIS LAUNCH uses for model m -- a kernel argument as in the reference: the models' own T_mc / T_cm are
 * not changed. sdfFrames / sdfs / nSdfs / normalThresholds / dDebugNorm are accepted and not read (the reference
 * kernel does not read the last two either, SURVEY A-20). Blocking (small device->host copies of the tables). */
int dartk_assoc_multi_tables(const float *d_vert, const float *d_norm, int width, int height, int nModels, const float *d_T_mcs,
                             const void *const *d_T_fms, const void *const *d_sdfFrames, const void *const *d_sdfs,
                             const int32_t *d_nSdfs, const float *d_distanceThresholds, const float *d_normalThresholds,
                             const float *d_planeOffsets, const float *d_planeNormals3, int32_t *d_lastElements,
                             dartk_da_point *const *d_pts, int32_t *d_dbgDA, float *d_dbgErr, float *d_dbgNorm, void *stream);
/* normEqnsObsToMod, obsToMod.h:12-21 / obsToMod.cu:621-697 (system zeroed; kernel only if nElements > 10). */
int dartk_ne_obs2mod(int dims, const float *d_vert, int width, int height, dartk_model_t m, float huberDelta,
                     const dartk_da_point *d_pts, int nElements, float *d_result, void *stream);
/* NO COUNTERPART IN THE REFERENCE. dartk_ne_obs2mod for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): same arguments, same kernels, same result, but the partial-system scratch
 * lives in a fixed device array inside the library instead of its hipMalloc'ed workspace, and the call performs no hipMalloc /
 * hipFree / allocating library call. dartk_init(device) must have run on the current device (it resolves the array's device
 * address; the entry point refuses a device it has not seen rather than initialise anything inside the call). Limits: dims <= 32 (the array holds 256 partial systems of that size, 561 KiB per GPU,
 * kept until the process ends); larger models are rejected with DARTK_ERR_INVALID. All calls on one GPU share the array: calls
 * that overlap on one GPU (two streams, two threads) race -- serialise them. */
int dartk_ne_obs2mod_noalloc(int dims, const float *d_vert, int width, int height, dartk_model_t m, float huberDelta,
                             const dartk_da_point *d_pts, int nElements, float *d_result, void *stream);
/* normEqnsObsToModReduced, obsToMod.h:23-33 / obsToMod.cu:699-757. d_dtheta_dalpha: [(dims-6)*(redDims-6)],
 * row = full joint (pose.cpp:66-73). */
int dartk_ne_obs2mod_reduced(int dims, int redDims, const float *d_dtheta_dalpha, const float *d_vert, int width,
                             int height, dartk_model_t m, float huberDelta, const dartk_da_point *d_pts,
                             int nElements, float *d_result, void *stream);
/* normEqnsObsToModParamMap, obsToMod.h:35-45 / obsToMod.cu:759-817. d_mapping: int[dims-6]. */
int dartk_ne_obs2mod_parammap(int dims, int redDims, const int32_t *d_mapping, const float *d_vert, int width,
                              int height, dartk_model_t m, float huberDelta, const dartk_da_point *d_pts,
                              int nElements, float *d_result, void *stream);

/* ------------------------------------------------------------------ mod -> obs (kernels/modToObs.h, raycast.h) */
/* raycastPrediction, raycast.cu:440-466 (one call per model; modelNum 0 clears the map). */
int dartk_raycast_prediction(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                             dartk_model_t m, float *d_prediction, float levelSet, void *stream);
/* raycastPrediction with the reference's own argument list (raycast.h:9-22; call-site prediction_renderer.cpp:195-207): T_mc by
 * value (host), the model's device arrays from the getters above. The handle is recovered from T_fms. */
int dartk_raycast_prediction_tables(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                                    const float *h_T_mc, const void *d_T_fms, const void *d_T_mfs, const void *d_sdfFrames,
                                    const void *d_sdfs, int nSdfs, float *d_prediction, float levelSet, void *stream);
/* cullUnobservable_, modToObs.cu:1050-1064 */
int dartk_cull_unobservable(float *d_pred, int predWidth, int predHeight, const float *d_obsVert, int obsWidth,
                            int obsHeight, void *stream);
/* Optimizer::generateObsSdfSplatZeroAndDistanceTransform, optimizer.cpp:168-219 = splatObsSdfZeros
 * (modToObs.cu:990-1008) + distanceTransform3D<float,true> (distance_transforms.cu:434-458) + copy back.
 * Uses the model's current T_cm. Any obs-SDF dims: the reference's splat grid has no ceil and needs multiples of (8,8,4)
 * (modToObs.cu:998-999); the grids here are ceil-divided. */
int dartk_obs_sdf_build(const float *d_vert, int width, int height, dartk_model_t m, float focalLength,
                        void *stream);
/* splatObsSdfZeros alone, modToObs.h:68-74 / modToObs.cu:990-1008 (kernel :403-441): writes 0 (behind the observed surface) /
 * 500 (no observation) / 1000 (free or outside the image) into the model's observation SDF (dartk_model_obs_sdf_data) for the
 * given model->camera transform h_T_cm (host SE3, by reference in the reference; the model's pose is not changed). The
 * reference's caller follows it with distanceTransform3D<float,true> into a scratch volume and a copy back
 * (optimizer.cpp:205-215): dartk_distance_transform_3d. dartk_obs_sdf_build does all three in one call. */
int dartk_splat_obs_sdf_zeros(const float *d_vert, int width, int height, const float *h_T_cm, dartk_model_t m, float focalLength,
                              void *stream);
/* distanceTransform3D / signedDistanceTransform3D on device buffers (distance_transforms.h:16-29);
 * d_in is clobbered as in the reference. */
int dartk_distance_transform_3d(float *d_in, float *d_out, int width, int height, int depth, int takeSqrt,
                                void *stream);
int dartk_signed_distance_transform_3d(float *d_in, float *d_out, int width, int height, int depth,
                                       int takeSqrt, void *stream);
/* normEqnsModToObs, modToObs.h:13-30 / modToObs.cu:713-780. h_T_gc = camera -> obs-SDF frame. Zeroes the
 * system and *d_numPredictions. */
int dartk_ne_mod2obs(int dims, const float *d_pred, int width, int height, dartk_model_t m, const float *h_T_gc,
                     float *d_result, int32_t *d_numPredictions, void *stream);
/* normEqnsModToObsTruncated, modToObs.cu:782-849 (uses the model's T_mc) */
int dartk_ne_mod2obs_truncated(int dims, const float *d_pred, int width, int height, dartk_model_t m,
                               float truncationDistance, float *d_result, int32_t *d_numPredictions,
                               void *stream);
/* normEqnsModToObsReduced / ParamMap, modToObs.cu:851-988 (use the model's T_mc; counter NOT zeroed, A-19) */
int dartk_ne_mod2obs_reduced(int dims, int redDims, const float *d_dtheta_dalpha, const float *d_pred, int width,
                             int height, dartk_model_t m, float *d_result, int32_t *d_numPredictions,
                             void *stream);
int dartk_ne_mod2obs_parammap(int dims, int redDims, const int32_t *d_mapping, const float *d_pred, int width,
                              int height, dartk_model_t m, float *d_result, int32_t *d_numPredictions,
                              void *stream);
/* The optional debug maps of normEqnsModToObs / Truncated / Reduced / ParamMap (modToObs.h:26-28,39-41,52-54,64-66; written at
 * modToObs.cu:47-49,65,74,82 and :145-147,168,178,186; the optimizer passes them when OptimizationOptions::debugModToObsDA / Err /
 * Norm are set, optimizer.cpp:296-324), by a kernel of their own so that the normal-equation kernels carry no stores for them:
 * d_dbgDA [height*width] label of the pixel's prediction when it is this model's (cleared to -1 only by the model with ID 0, as
 * there), d_dbgErr [height*width] its residual (cleared to NaN only by model 0), d_dbgNorm [height*width][4] the observation
 * SDF's gradient, w = 1 (always cleared to 0). Any of the three may be NULL. h_T_gc: the transform normEqnsModToObs is given, or
 * NULL for the model's own camera->model transform (what the other three wrappers use). truncated != 0: the Truncated
 * kernel's rule (pixels outside [0, truncationDistance) leave no residual). */
int dartk_mod2obs_debug_maps(const float *d_pred, int width, int height, dartk_model_t m, const float *h_T_gc, int truncated,
                             float truncationDistance, int32_t *d_dbgDA, float *d_dbgErr, float *d_dbgNorm, void *stream);
/* errorModToObs, modToObs.cu:1030-1048: returns 0.5*sum err^2 in *h_error */
int dartk_error_mod2obs(const float *d_pred, int width, int height, dartk_model_t m, float *h_error,
                        void *stream);

/* ------------------------------------------------------------------ intersection (kernels/intersection.h) */
/* normEqnsSelfIntersection{,Reduced,ParamMap}, intersection.cu:765-890. Result is the (dims-6) system. */
int dartk_ne_self_intersection(const float *d_sites, int nSites, int dims, dartk_model_t m,
                               const int32_t *d_potentialIntersection, float *d_result, void *stream);
int dartk_ne_self_intersection_reduced(const float *d_sites, int nSites, int fullDims, int redDims,
                                       dartk_model_t m, const float *d_dtheta_dalpha,
                                       const int32_t *d_potentialIntersection, float *d_result, void *stream);
int dartk_ne_self_intersection_parammap(const float *d_sites, int nSites, int fullDims, int redDims,
                                        dartk_model_t m, const int32_t *d_mapping,
                                        const int32_t *d_potentialIntersection, float *d_result, void *stream);
/* normEqnsIntersection{,Reduced,ParamMap}, intersection.cu:792-947. T_ds/T_sd host SE3 (optimizer.cpp:396-397). */
int dartk_ne_intersection(const float *d_sites, int nSites, int dims, const float *h_T_ds, const float *h_T_sd,
                          dartk_model_t src, dartk_model_t dst, float *d_result, void *stream);
int dartk_ne_intersection_reduced(const float *d_sites, int nSites, int fullDims, int redDims,
                                  const float *h_T_ds, const float *h_T_sd, dartk_model_t src, dartk_model_t dst,
                                  const float *d_dtheta_dalpha, float *d_result, void *stream);
int dartk_ne_intersection_parammap(const float *d_sites, int nSites, int fullDims, int redDims,
                                   const float *h_T_ds, const float *h_T_sd, dartk_model_t src, dartk_model_t dst,
                                   const int32_t *d_mapping, float *d_result, void *stream);
/* The terms' optional debugError output (intersection.h:26,54,78-79; the optimizer passes one array over all models' collision
 * clouds when OptimizationOptions::debugIntersectionErr is set, optimizer.cpp:646-682): initDebugIntersectionError sets NaN
 * (intersection.cu:303-313, 988-995), a colliding site of the self term gets residual^2 (:131-133), of a pairwise term
 * += residual^2 (:433-435: NaN stays NaN unless the self term wrote the site first); other sites are left as they are.
 * Kernels of their own, same collision test as the normal-equation kernels. */
int dartk_init_debug_intersection_error(float *d_debugError, int nSites, void *stream);
int dartk_debug_self_intersection_error(const float *d_sites, int nSites, dartk_model_t m,
                                        const int32_t *d_potentialIntersection, float *d_debugError, void *stream);
int dartk_debug_intersection_error(const float *d_sites, int nSites, const float *h_T_ds, dartk_model_t src,
                                   dartk_model_t dst, float *d_debugError, void *stream);
/* countSelfIntersections / countIntersections, intersection.cu:715-763 (blocking; return the count) */
int dartk_count_self_intersections(const float *d_sites, int nSites, dartk_model_t m,
                                   const int32_t *d_potentialIntersection, int32_t *h_count, void *stream);
int dartk_count_intersections(const float *d_sites, int nSites, const float *h_T_ds, dartk_model_t src,
                              dartk_model_t dst, int32_t *h_count, void *stream);
/* countSelfIntersections with the reference's own argument list (intersection.h:11-18): raw device arrays, handle recovered
 * from T_mfs; blocking, returns the count in *h_count. */
int dartk_count_self_intersections_tables(const float *d_sites, int nSites, const void *d_T_mfs, const void *d_T_fms,
                                          const void *d_sdfFrames, const void *d_sdfs, int nSdfs,
                                          const int32_t *d_potentialIntersection, int32_t *h_count);

/* ---- model / asset pipeline: solid-mesh voxel splat (mesh/mesh_splat.cpp:10-265, splatSolidMesh) -------------------
 * Sets d_sdf[x + y*dimX + z*dimX*dimY] = 0 for every voxel that at least two of the three axis-parallel parity ray
 * casts find inside the triangle mesh (d_vertices: float[3*nVertices], d_faces: int32[3*nFaces], both in the grid's
 * coordinate frame, i.e. after scaleMesh/transformMesh, host_only_model.cpp:813-826); other voxels are left as they
 * are (the caller presets the background, 1e20 in HostOnlyModel::voxelize, host_only_model.cpp:290). Synchronises
 * the stream before returning (its scratch is released). */
int dartk_splat_solid_mesh(const float *d_vertices, int nVertices, const int32_t *d_faces, int nFaces, int dimX, int dimY,
                           int dimZ, const float *h_offset /*[3]*/, float resolution, float *d_sdf, void *stream);

/* ------------------------------------------------------------------ synthetic depth front end */
/* depthToVertices<float> (img_proc/organized_point_cloud.h; .cu:11-40, :381-395): back-projection
 * ((u-pp.x)*z/f.x, (v-pp.y)*z/f.y, z, 1) for range.x <= z <= range.y, else w = 0. */
int dartk_depth_to_vertices(const float *d_depth, float *d_vert, int width, int height, float ppx, float ppy,
                            float flx, float fly, float rangeMin, float rangeMax, void *stream);
/* verticesToNormals (organized_point_cloud.cu:265-305, :437-449) */
int dartk_vertices_to_normals(const float *d_vert, float *d_norm, int width, int height, void *stream);
/* No counterpart in the reference (it reads sensors / renders with OpenGL): depth image = z channel of a
 * ray-cast prediction map, the on-device synthetic depth source north_star asks for. */
int dartk_prediction_to_depth(const float *d_pred, float *d_depth, int width, int height, void *stream);

/* ------------------------------------------------------------------ device-resident Gauss-Newton loops */
/* OptimizationOptions (optimization.h:9-72) for one model + the PredictionRenderer geometry the Optimizer
 * constructor derives (optimizer.cpp:72-98: prediction size and focal length scaled by predWidth/depthWidth). */
typedef struct {
    int32_t numIterations;
    float distThreshold, normThreshold, planeOffset;
    float planeNormal[3];
    float regularization, regularizationScaled;
    float lambdaObsToMod, lambdaModToObs, lambdaIntersection;
    float huberDelta, focalLength;
    int32_t predWidth, predHeight;
    float predFl[2];
} dartk_opt_options;

/* Optimizer::optimizePose (optimizer.h:25-35 / optimizer.cpp:423-522) for a NullReduction pose, entirely on the
 * device: obs-SDF build (if lambdaModToObs>0), then numIterations x { ray-cast prediction, association + obs->mod
 * normal equations, mod->obs normal equations, optional self-intersection, assemble (unpack, :50-70), A = JTJ +
 * reg*I, A_ii += lm*A_ii, LDL^T solve, T_mc <- exp(d)*T_mc, joints clamped, forward kinematics }. The pose is the
 * model's current pose (dartk_model_set_pose) and is updated in place; read it back with
 * dartk_model_get_transforms. d_prediction: float4[predWidth*predHeight] scratch. Optional outputs:
 * d_iterInfo[8*numIterations] = {errObsToMod, errModToObs, errIntersection, nAssociated, nPredicted,0,0,0} per
 * iteration (Optimizer::_iterationSummaries); d_firstSystem = the first iteration's three packed systems. */
int dartk_optimize_pose(dartk_model_t m, const float *d_vert, const float *d_norm, int width, int height,
                        const dartk_opt_options *opts, const float *d_collisionCloud, int nCloud,
                        const int32_t *d_potentialIntersection, float *d_prediction, float *d_iterInfo,
                        float *d_firstSystem, void *stream);

/* Optimizer::optimizePose (optimizer.cpp:423-522) for nFrames INDEPENDENT frames of one model in one call (BASELINE configs[1]
 * over a recorded sequence: the reference loops over the frames on the host, one optimizePose each). Every launch covers all
 * frames (grid.y = frame; the LDL^T solve: one wave per frame) with the kernels of dartk_optimize_pose and, per frame, the same
 * partial systems summed in the same order: frame f's result equals, bit for bit, dartk_optimize_pose on a model set to
 * (d_T_mcs[f], d_joints[f]) -- no atomically appended list is involved on this path. The frames share the model's SDFs and
 * tables, the options, the collision cloud and the potential matrix, and differ in
 *   d_vert / d_norm   [nFrames][height*width*4]  observation maps, frame-major
 *   d_T_mcs           [nFrames][12]              camera->model transforms, in/out
 *   d_joints          [nFrames][nJoints]         in/out (NULL for a rigid model)
 * Optional d_iterInfo [nFrames][numIterations][8] as for dartk_optimize_pose. The model handle's own pose and observation SDF
 * are not touched. Frames are processed in chunks of 64 (DARTK_BATCH_CHUNK), which bounds the device memory held for the
 * frames' partial systems, traces, prediction maps and observation SDFs. Models of more than 128 SDFs: DARTK_ERR_UNSUPPORTED
 * when the mod->obs term is on. */
int dartk_optimize_pose_batch(dartk_model_t m, int nFrames, const float *d_vert, const float *d_norm, int width, int height,
                              const dartk_opt_options *opts, const float *d_collisionCloud, int nCloud,
                              const int32_t *d_potentialIntersection, float *d_T_mcs, float *d_joints, float *d_iterInfo,
                              void *stream);

/* ------------------------------------------------------------------ ContactGrasp refinement (TrackerNoObs / OptimizerNoObs) */
/* ContactPrior (optimization/priors.h:35-91): contact point in the source SDF's frame, attracted to (or, inverted,
 * repelled from) the nearest SDF of the destination model. */
typedef struct {
    int32_t srcModel, dstModel, srcSdf;   /* _srcModelID, _dstModelID, _srcSdfNum (dst SDF: nearest, reset every evaluation, optimizer.cpp:914) */
    float weight;                         /* _weight */
    float contactPoint[3];                /* _contactPoint (source SDF frame) */
    float normal[3];                      /* _normal_sf */
    int32_t invert;                       /* repulsive */
    float contactThreshold;               /* _contactThreshold */
    int32_t updateSrc, updateDst, updateContact;   /* _update_src_pose, _update_dst_pose, _update_contact_point */
} dartk_contact_prior;

/* The OptimizationOptions fields OptimizerNoObs reads (optimization.h:9-72), per model. */
typedef struct {
    int32_t numIterations, nModels;
    float regularization[DARTK_MAX_MODELS];
    float regularizationScaled[DARTK_MAX_MODELS];
    float lambdaIntersection[DARTK_MAX_MODELS * DARTK_MAX_MODELS];   /* [src + dst*nModels], optimizer.cpp:866,873 */
} dartk_noobs_options;

typedef struct dartk_noobs_s *dartk_noobs_t;

/* TrackerNoObs state for one set of models (tracker_no_obs.cpp:58-235, 271-299): collision clouds (float4, xyz in the
 * owning SDF's frame, w = SDF number), intersection-potential matrices, contact priors, and a pose / partial-system
 * workspace for up to maxBatch independent candidates. Model m has model ID m. The intersection terms that exist
 * (lambdaIntersection > 0) are fixed here; their values may change per call. Damping matrices start as zero
 * (tracker_no_obs.cpp:214-215; dartk_noobs_set_damping). */
int dartk_noobs_create(int nModels, const dartk_model_t *models, const float *const *d_collisionClouds,
                       const int32_t *nCloudPoints, const int32_t *const *d_potentialIntersection,
                       const float *h_lambdaIntersection, const dartk_contact_prior *h_priors, int nPriors, int maxBatch,
                       dartk_noobs_t *out);
/* The same with pose reductions (pose.h: NullReduction / LinearPoseReduction / ParamMapPoseReduction; the reference's Barrett
 * ContactGrasp set-up, grasp_analyzer.cpp:393-420, refines a LinearPoseReduction(8 -> 4) pose): OptimizerNoObs works in the
 * REDUCED dimensions 6 + redArtDims per model (optimizer.cpp:821-823), the intersection terms use their Reduced / ParamMap forms
 * (optimizer.cpp:332-421), the update clamps the reduced articulation to the reduction's limits and projects it to the joints
 * (optimizer.cpp:945-952). h_reductions: [nModels] or NULL (no model reduced). For a reduced model the joint arrays of
 * dartk_noobs_optimize_poses / _get_error carry its redArtDims reduced values instead of its joints. */
typedef struct dartk_reduction {
    int32_t type;                       /* 0 null, 1 linear (full = b + A*reduced, pose.cpp:66-73), 2 parameter map (pose.cpp:91-97) */
    int32_t redArtDims;                 /* ignored for type 0 */
    const float *h_A;                   /* types 1, 2: first derivatives [nJoints][redArtDims] (PoseReduction::getFirstDerivatives) */
    const int32_t *h_map;               /* type 2: [nJoints] */
    const float *h_b;                   /* type 1: [nJoints] offsets, NULL = zeros */
    const float *h_redMin, *h_redMax;   /* [redArtDims] */
} dartk_reduction;
int dartk_noobs_create_reduced(int nModels, const dartk_model_t *models, const dartk_reduction *h_reductions,
                               const float *const *d_collisionClouds, const int32_t *nCloudPoints,
                               const int32_t *const *d_potentialIntersection, const float *h_lambdaIntersection,
                               const dartk_contact_prior *h_priors, int nPriors, int maxBatch, dartk_noobs_t *out);
int dartk_noobs_destroy(dartk_noobs_t t);
/* TrackerNoObs::getDampingMatrix (tracker_no_obs.h:73): the matrix OptimizerNoObs adds to model modelNum's block before the
 * priors contribute (optimizer.cpp:895-896). h_matrix: [dims][dims] row-major, dims = the model's reduced dimensions
 * (6 + joints, or 6 + redArtDims); NULL = zero again. Shared by all candidates of a batch. */
int dartk_noobs_set_damping(dartk_noobs_t t, int modelNum, const float *h_matrix, void *stream);
/* ContactPrior::setWeight / setContactThreshold / contact point resets (priors.h:66-67, grasp_analyzer.cpp:941-981) */
int dartk_noobs_set_priors(dartk_noobs_t t, const dartk_contact_prior *h_priors, int nPriors, void *stream);
/* Point3D3DPrior / Point2D3DPrior (src/optimization/priors.h:92-170; computeContribution point_3d_3d_prior.cpp:6-69,
 * point_2d_3d_prior.cpp:6-83): a point fixed in a frame of the source model is pulled to a camera-frame point (kind 0:
 * residual = R_mc (p_c - target), 3 rows) or to a pixel (kind 1: residual = projection - target, 2 rows). They add
 * w J^T J and w J^T e to the source model's block and nothing to the energy (their _error stays 0), but count in the
 * energy's division by priors.size() (optimizer.cpp:1068). */
typedef struct dartk_point_prior {
    int32_t kind;               /* 0 Point3D3DPrior, 1 Point2D3DPrior */
    int32_t srcModel, srcFrame;
    float weight;
    float point_f[3];           /* _point_f: the point in the source frame */
    float point_c[3];           /* _point_c: target, camera frame (kind 0) or pixel x, y (kind 1; [2] unused) */
    float focalLength[2], principalPoint[2];    /* kind 1 */
} dartk_point_prior;
/* Replaces the tracker's list of point priors (TrackerNoObs::addPrior, tracker_no_obs.h:47); evaluated after the contact
 * priors. A set-up call: synchronises, and re-allocates the partial-system buffer when n changes. */
int dartk_noobs_set_point_priors(dartk_noobs_t t, const dartk_point_prior *h_priors, int nPriors, void *stream);

/* OptimizerNoObs::optimizePoses (optimizer.h:141-160 / optimizer.cpp:783-983) for `batch` independent candidates
 * (NullReduction poses): per iteration self- and pairwise intersection terms, every contact prior
 * (ContactPrior::computeContribution, contact_prior.cpp:8-205), regularisation + LM damping after the priors, the
 * upper-triangular back substitution, pose update; then getError (optimizer.cpp:985-1076) -> d_energy[batch].
 * d_T_mcs: [batch][nModels][12] camera->model, in/out. d_joints: [batch][sum of joints] in/out (model order).
 * Optional d_iterInfo [batch][numIterations][8] = {intersection e, sum of prior _error, energy, 0...};
 * d_firstSystem [batch][D*D + D] = upper JTJ and JTe of iteration 0 before regularisation, D = sum of model dims. */
int dartk_noobs_optimize_poses(dartk_noobs_t t, const dartk_noobs_options *opts, int batch, float *d_T_mcs,
                               float *d_joints, float *d_energy, float *d_iterInfo, float *d_firstSystem, void *stream);
/* Contact priors with update_contact_point (contact_prior.cpp:104-108,151-186): the prior's three contact-point unknowns
 * are solved with the system (optimizer.cpp:829-832,931), ContactPrior::updatePriorParams (contact_prior.cpp:207-221) moves
 * the contact point by its share of the update and projectToSdfSurface (sdf.cpp:6-23) puts it back onto its SDF's surface.
 * The reference mutates the prior object; here every candidate of the batch carries its own copy of the contact points,
 * started from the priors' at each dartk_noobs_optimize_poses call. This reads them back after that call:
 * d_contactPoints [batch][nPriors][3] (source-SDF frame). */
int dartk_noobs_contact_points(dartk_noobs_t t, int batch, float *d_contactPoints, void *stream);
/* OptimizerNoObs::getError (optimizer.cpp:985-1076) at the given poses (and the priors' own contact points) */
int dartk_noobs_get_error(dartk_noobs_t t, const dartk_noobs_options *opts, int batch, const float *d_T_mcs,
                          const float *d_joints, float *d_energy, void *stream);

/* ------------------------------------------------------------------ multi-model tracking loop on the device */
/* One model of Optimizer::optimizePoses (optimizer.cpp:524-781): its device mirror (pose already set and consistent with
 * d_redArt), its pose reduction (pose.h: NullReduction / LinearPoseReduction / ParamMapPoseReduction), collision cloud,
 * potential-intersection matrix, association list and per-model options (optimization.h:30-70). */
typedef struct dartk_multi_model {
    dartk_model_t model;
    int32_t redType;            /* 0 null, 1 linear (full = b + A*reduced, pose.cpp:66-73), 2 parameter map (pose.cpp:91-97) */
    int32_t redArtDims;         /* reduced articulation dimensions (ignored for type 0: the joint count) */
    const float *d_A;           /* type 1: first derivatives [nJoints][redArtDims], device */
    const int32_t *d_map;       /* type 2: mapping [nJoints], device */
    const float *h_b;           /* type 1: offsets [nJoints], host (NULL = zeros) */
    const float *h_redMin, *h_redMax;   /* [redArtDims] limits of the reduced articulation, host */
    float *d_redArt;            /* [redArtDims] reduced articulation, device, in/out */
    const float *d_cloud;       /* collision cloud [nCloud][4] (frame coordinates, .w = SDF number) or NULL */
    int32_t nCloud;
    const int32_t *d_potential; /* [nSdfs*nSdfs] or NULL */
    dartk_da_point *d_pts;      /* association list, capacity width*height */
    float distThreshold, planeOffset, planeNormal[3];
    float regularization, regularizationScaled;
    /* Tracker::getDampingMatrix (tracker.h:122; optimizer.cpp:702-703: added to the model's block after the regularisation and
     * the LM scaling): [dims][dims] row-major, dims = 6 + redArtDims, device, or NULL (zero, the tracker's initial value) */
    const float *d_damping;
} dartk_multi_model;
typedef struct dartk_multi_options {
    int32_t numIterations, nModels;
    float lambdaObsToMod, lambdaModToObs, huberDelta, focalLength;
    int32_t predWidth, predHeight;
    float predFl[2];
    float lambdaIntersection[DARTK_MAX_MODELS * DARTK_MAX_MODELS];      /* [src + dst*nModels] */
} dartk_multi_options;
/* Optimizer::optimizePoses (optimizer.h:86-110 / optimizer.cpp:524-781) without priors, device-resident: per iteration the
 * ray-cast of every model + cullUnobservable, the multi-model association, per model obs->mod / mod->obs / self- and
 * pairwise intersection normal equations in the pose's reduced dimensions, the block assembly with regularisation and LM
 * scaling, the upper-triangular back substitution, the reduced pose updates with their limits, the projection to full joints
 * and forward kinematics. The models' device poses and d_redArt hold the result. Optional d_iterInfo
 * [numIterations][nModels][8] = {lambda*e obs->mod, mod->obs, last intersection term, #associated, #predicted, 0..};
 * d_firstSystem [D*D + D] = upper JTJ and JTe of iteration 0 after regularisation, D = sum of reduced dims (<= 64). */
int dartk_optimize_poses(const dartk_multi_options *opts, const dartk_multi_model *models, const float *d_vert,
                         const float *d_norm, int width, int height, float *d_prediction, float *d_iterInfo,
                         float *d_firstSystem, void *stream);

/* The same with the tracker's contact priors (Tracker::addPrior; optimizer.cpp:727-732: after the model blocks are assembled every
 * prior adds its contribution, ContactPrior::computeContribution, contact_prior.cpp:8-205, and :757-759: after the pose update
 * every prior takes its share of the parameter update, ContactPrior::updatePriorParams, contact_prior.cpp:207-221, with
 * projectToSdfSurface, sdf.cpp:6-23). The tracking loop carries a handful of priors (hand-object contacts): they are evaluated
 * inside the solve kernel, one after the other. h_priors [nPriors] is uploaded to d_priors [nPriors] (device, caller-owned)
 * at the start of the call; priors with updateContact leave their moved contact point in d_priors[p].contactPoint. */
int dartk_optimize_poses_priors(const dartk_multi_options *opts, const dartk_multi_model *models, dartk_contact_prior *d_priors,
                                const dartk_contact_prior *h_priors, int nPriors, const float *d_vert, const float *d_norm,
                                int width, int height, float *d_prediction, float *d_iterInfo, float *d_firstSystem, void *stream);

/* Optimizer::optimizePoses (optimizer.cpp:524-781) for nFrames INDEPENDENT frames in one call (BASELINE configs[4]: a batch of
 * recorded frames tracked from their own start poses; the reference would loop over the frames on the host). Every kernel launch
 * covers all frames (grid.y = frame, the solve one wave per frame): the frames share the models' SDFs, tables, reductions,
 * clouds and options (`models`, as for dartk_optimize_poses; its d_redArt / d_pts fields are not used here) and differ in
 *   d_vert / d_norm   [nFrames][height*width*4]  observation maps, frame-major
 *   d_T_mcs           [nFrames][nModels][12]     camera->model transforms, in/out
 *   d_redArt          [nFrames][sum of reduced articulation dims, model order]  in/out (NULL when no model is articulated)
 * Optional d_iterInfo [nFrames][numIterations][nModels][8] as for dartk_optimize_poses. The model handles' own poses are not
 * touched. Frames are processed in chunks of 64 (DARTK_BATCH_CHUNK) to bound the device memory held for association lists,
 * ray-cast traces and prediction maps. Per frame the result equals a dartk_optimize_poses call on that frame (same kernels;
 * sums of the atomically appended association lists differ in order, as between two single calls). */
int dartk_optimize_poses_batch(const dartk_multi_options *opts, const dartk_multi_model *models, int nFrames, const float *d_vert,
                               const float *d_norm, int width, int height, float *d_T_mcs, float *d_redArt, float *d_iterInfo,
                               void *stream);

/* ------------------------------------------------------------------ measurement (no reference counterpart) */
/* Per-kernel-class device time of the Gauss-Newton loops, measured with HIP events recorded on the launch stream
 * around every launch between dartk_profile_begin() and dartk_profile_end(). Used by bench.py's roofline leg;
 * adds two event records per launch, so throughput is timed with profiling off. */
enum dartk_prof_class {
    DARTK_PROF_OBS_SDF = 0, DARTK_PROF_RAYCAST = 1 /* trace kernel */, DARTK_PROF_FUSED_OBS2MOD = 2, DARTK_PROF_NE_MOD2OBS = 3,
    DARTK_PROF_NE_SELF = 4, DARTK_PROF_SOLVE = 5, DARTK_PROF_RAYCAST_REPLAY = 6, DARTK_PROF_OTHER = 7, DARTK_PROF_REDUCE = 8,
    DARTK_PROF_NOOBS_TERMS = 9 /* self- and pairwise intersection term kernels */, DARTK_PROF_NOOBS_SOLVE = 10,
    DARTK_PROF_NOOBS_PRIORS = 11 /* contact-prior term kernel */, DARTK_PROF_ASSOC_MULTI = 12 /* multi-model association */,
    DARTK_PROF_CLASSES = 13
};
/* Trilinear SDF lookups actually performed by the kernels since the last reset, by category: h_counts[8] = {other, association
 * scan, association gradient (normal test / fused Jacobian), obs->mod gradient, ray-cast march, mod->obs (value + gradient),
 * intersection, contact priors}. Only the diagnostic build libdartk_count.so (-DDARTK_COUNT_LOOKUPS, one atomic per lookup)
 * counts; libdartk.so returns DARTK_ERR_UNSUPPORTED. Model handles are interchangeable between the two libraries. */
int dartk_lookup_counters(uint64_t *h_counts, int reset);
int dartk_profile_begin(void);
int dartk_profile_end(float *h_ms_total /*[DARTK_PROF_CLASSES]*/, int32_t *h_launches /*[DARTK_PROF_CLASSES]*/);

#ifdef __cplusplus
}
#endif
#endif /* DARTK_H */
