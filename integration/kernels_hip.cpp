// kernels_hip.cpp -- the translation unit a DART maintainer compiles INSTEAD OF the reference's CUDA kernel files: every
// `namespace dart` free function declared in
//     src/optimization/kernels/obsToMod.h:12-107      src/optimization/kernels/modToObs.h:13-90
//     src/optimization/kernels/intersection.h:11-79   src/optimization/kernels/raycast.h:9-22
// (+ the two geometry / img_proc entry points the optimizer and the point-cloud source call), with the reference's own parameter
// lists, forwarding to the C ABI of libdartk.so (include/dartk.h). Host code above it -- optimizer.cpp, prediction_renderer.cpp,
// tracker*.cpp -- stays as it is. integration/check_signatures.py diffs the parameter lists below against the text of the
// reference's headers; tests/cpp/stub_kat1.cpp runs the reference's kernel test through dart::errorAndDataAssociation.
//
// The one spelling that differs: cudaStream_t -> hipStream_t (the host code is built against HIP). Error handling follows
// CheckCudaDieOnError (util/cuda_utils.h:7-17): print and exit(1).
// Not defined here, on purpose: normEqnsObsToModAndPinkyPos (declared at obsToMod.h:47-73, defined nowhere in the reference),
// getErrorJacobianOfModelPoint (a __device__ function), computeTruncatedObsSdf (its only caller is commented out,
// optimizer.cpp:163-164) and raycastPredictionDebugRay (a GUI debugging aid, prediction_renderer.cpp:320).
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "dart_compat.h"

namespace dart {

namespace {

void dieOnError(int rc, const char *what) {
    if (rc == 0) return;
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, dartk_last_error());
    exit(1);
}
#define DARTK_CALL(call) dieOnError((call), #call)

const float *f(const float4 *p) { return reinterpret_cast<const float *>(p); }
float *f(float4 *p) { return reinterpret_cast<float *>(p); }
const float *f(const SE3 &T) { return reinterpret_cast<const float *>(&T); }
dartk_da_point *da(DataAssociatedPoint *p) { return reinterpret_cast<dartk_da_point *>(p); }

// the model that owns a device array handed out by MirroredModel's getters (the observation SDF, a transform table, ...)
dartk_model_t owner(const void *devicePtr, const char *what) {
    dartk_model_t m = dartk_model_from_device_pointer(devicePtr);
    if (!m) { fprintf(stderr, "%s: not a device array of a live MirroredModel\n", what); exit(1); }
    return m;
}

// a device counter for callers that pass numPredictions = 0 (its default in modToObs.h:25,63)
int *spareCounter() {
    static int *d = 0;
    if (!d) { void *p = 0; DARTK_CALL(dartk_buf_alloc(&p, 4 * sizeof(int))); d = static_cast<int *>(p); }
    return d;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ obsToMod.h
void normEqnsObsToMod(const int dims,
                      const float4 * dObsVertMap,
                      const int width,
                      const int height,
                      const MirroredModel & model,
                      const OptimizationOptions & opts,
                      DataAssociatedPoint * dPts,
                      int nElements,
                      float * dResult,
                      float4 * debugJs) {
    (void)debugJs;      // no caller passes it (optimizer.cpp:261-269 leaves the default 0)
    DARTK_CALL(dartk_ne_obs2mod(dims, f(dObsVertMap), width, height, model.dartk(), opts.huberDelta, da(dPts), nElements, dResult, 0));
}

void normEqnsObsToModReduced(const int dims,
                             const int reductionDims,
                             const float * d_dtheta_dalpha,
                             const float4 * dObsVertMap,
                             const int width,
                             const int height,
                             const MirroredModel & model,
                             const OptimizationOptions & opts,
                             DataAssociatedPoint * dPts,
                             int nElements,
                             float * dResult) {
    DARTK_CALL(dartk_ne_obs2mod_reduced(dims, reductionDims, d_dtheta_dalpha, f(dObsVertMap), width, height, model.dartk(), opts.huberDelta,
                                        da(dPts), nElements, dResult, 0));
}

void normEqnsObsToModParamMap(const int dims,
                              const int reductionDims,
                              const int * dMapping,
                              const float4 * dObsVertMap,
                              const int width,
                              const int height,
                              const MirroredModel & model,
                              const OptimizationOptions & opts,
                              DataAssociatedPoint * dPts,
                              int nElements,
                              float * dResult) {
    DARTK_CALL(dartk_ne_obs2mod_parammap(dims, reductionDims, dMapping, f(dObsVertMap), width, height, model.dartk(), opts.huberDelta,
                                         da(dPts), nElements, dResult, 0));
}

void errorAndDataAssociation(const float4 * dObsVertMap,
                             const float4 * dObsNormMap,
                             const int width,
                             const int height,
                             const MirroredModel & model,
                             const OptimizationOptions & opts,
                             DataAssociatedPoint * dPts,
                             int * dLastElement,
                             int * hLastElement,
                             int * debugDataAssociation,
                             float * debugError,
                             float4 * debugNorm) {
    // single-model path: the options of model 0 (obsToMod.cu:514-551 reads opts.distThreshold[0], planeOffset[0], planeNormal[0])
    const float pn[3] = {opts.planeNormal[0].x, opts.planeNormal[0].y, opts.planeNormal[0].z};
    DARTK_CALL(dartk_assoc_single(f(dObsVertMap), f(dObsNormMap), width, height, model.dartk(), opts.distThreshold[0], opts.normThreshold,
                                  opts.planeOffset[0], pn, da(dPts), dLastElement, hLastElement, debugDataAssociation, debugError,
                                  f(debugNorm), 0));
}

void errorAndDataAssociationMultiModel(const float4 * dObsVertMap,
                                       const float4 * dObsNormMap,
                                       const int width,
                                       const int height,
                                       const int nModels,
                                       const SE3 * T_mcs,
                                       const SE3 * const * T_fms,
                                       const int * const * sdfFrames,
                                       const Grid3D<float> * const * sdfs,
                                       const int * nSdfs,
                                       const float * distanceThresholds,
                                       const float * normalThresholds,
                                       const float * planeOffsets,
                                       const float3 * planeNormals,
                                       int * lastElements,
                                       DataAssociatedPoint * * pts,
                                       int * dDebugDataAssociation,
                                       float * dDebugError,
                                       float4 * dDebugNorm,
                                       hipStream_t stream) {
    // every pointer is a device pointer, the per-model tables device arrays of device pointers (optimizer.cpp:620-630)
    DARTK_CALL(dartk_assoc_multi_tables(f(dObsVertMap), f(dObsNormMap), width, height, nModels, reinterpret_cast<const float *>(T_mcs),
                                        reinterpret_cast<const void *const *>(T_fms), reinterpret_cast<const void *const *>(sdfFrames),
                                        reinterpret_cast<const void *const *>(sdfs), nSdfs, distanceThresholds, normalThresholds,
                                        planeOffsets, reinterpret_cast<const float *>(planeNormals), lastElements,
                                        reinterpret_cast<dartk_da_point *const *>(pts), dDebugDataAssociation, dDebugError,
                                        f(dDebugNorm), stream));
}

// ------------------------------------------------------------------------------------------------ modToObs.h
float errorModToObs(const float4 * dLabeledPredictedVertMap,
                    const int width,
                    const int height,
                    const Grid3D<float> * dObsSdf) {
    float e = 0;
    DARTK_CALL(dartk_error_mod2obs(f(dLabeledPredictedVertMap), width, height, owner(dObsSdf, "errorModToObs"), &e, 0));
    return e;
}

void normEqnsModToObs(const int dimensions,
                      const float4 * dLabeledPredictedVertMap,
                      const int width,
                      const int height,
                      const MirroredModel & model,
                      const SE3 T_gc,
                      float * dResult,
                      int * numPredictions,
                      int * debugDataAssociation,
                      float * debugError,
                      float4 * debugNorm) {
    DARTK_CALL(dartk_ne_mod2obs(dimensions, f(dLabeledPredictedVertMap), width, height, model.dartk(), f(T_gc), dResult,
                                numPredictions ? numPredictions : spareCounter(), 0));
    if (debugDataAssociation || debugError || debugNorm)
        DARTK_CALL(dartk_mod2obs_debug_maps(f(dLabeledPredictedVertMap), width, height, model.dartk(), f(T_gc), 0, 0.0f,
                                            debugDataAssociation, debugError, f(debugNorm), 0));
}

void normEqnsModToObsReduced(const int dims,
                             const int reductionDims,
                             const float * d_dtheta_dalpha,
                             const float4 * dLabeledPredictedVertMap,
                             const int width,
                             const int height,
                             const MirroredModel & model,
                             float * dResult,
                             int * numPredictions,
                             int * debugDataAssociation,
                             float * debugError,
                             float4 * debugNorm) {
    DARTK_CALL(dartk_ne_mod2obs_reduced(dims, reductionDims, d_dtheta_dalpha, f(dLabeledPredictedVertMap), width, height, model.dartk(),
                                        dResult, numPredictions ? numPredictions : spareCounter(), 0));
    if (debugDataAssociation || debugError || debugNorm)
        DARTK_CALL(dartk_mod2obs_debug_maps(f(dLabeledPredictedVertMap), width, height, model.dartk(), 0, 0, 0.0f, debugDataAssociation,
                                            debugError, f(debugNorm), 0));
}

void normEqnsModToObsParamMap(const int dims,
                              const int reductionDims,
                              const int * dMapping,
                              const float4 * dLabeledPredictedVertMap,
                              const int width,
                              const int height,
                              const MirroredModel & model,
                              float * dResult,
                              int * numPredictions,
                              int * debugDataAssociation,
                              float * debugError,
                              float4 * debugNorm) {
    DARTK_CALL(dartk_ne_mod2obs_parammap(dims, reductionDims, dMapping, f(dLabeledPredictedVertMap), width, height, model.dartk(), dResult,
                                         numPredictions ? numPredictions : spareCounter(), 0));
    if (debugDataAssociation || debugError || debugNorm)
        DARTK_CALL(dartk_mod2obs_debug_maps(f(dLabeledPredictedVertMap), width, height, model.dartk(), 0, 0, 0.0f, debugDataAssociation,
                                            debugError, f(debugNorm), 0));
}

void normEqnsModToObsTruncated(const int dimensions,
                               const float4 * dLabeledPredictedVertMap,
                               const int width,
                               const int height,
                               const MirroredModel & model,
                               const float truncationDistance,
                               float * dResult,
                               int * numPredictions,
                               int * debugDataAssociation,
                               float * debugError,
                               float4 * debugNorm) {
    DARTK_CALL(dartk_ne_mod2obs_truncated(dimensions, f(dLabeledPredictedVertMap), width, height, model.dartk(), truncationDistance, dResult,
                                          numPredictions ? numPredictions : spareCounter(), 0));
    if (debugDataAssociation || debugError || debugNorm)
        DARTK_CALL(dartk_mod2obs_debug_maps(f(dLabeledPredictedVertMap), width, height, model.dartk(), 0, 1, truncationDistance,
                                            debugDataAssociation, debugError, f(debugNorm), 0));
}

void splatObsSdfZeros(const float4 * dObsVertMap,
                      const int width,
                      const int height,
                      const SE3 & T_cm,
                      const Grid3D<float> * dObsSdf,
                      const uint3 sdfDim,
                      const float focalLength) {
    (void)sdfDim;       // the model knows its observation SDF's size (the reference passes it only to shape the launch)
    DARTK_CALL(dartk_splat_obs_sdf_zeros(f(dObsVertMap), width, height, f(T_cm), owner(dObsSdf, "splatObsSdfZeros"), focalLength, 0));
}

void cullUnobservable_(float4 * predVertMap,
                      const int predWidth,
                      const int predHeight,
                      const float4 * obsVertMap,
                      const int obsWidth,
                      const int obsHeight,
                      const hipStream_t stream) {
    DARTK_CALL(dartk_cull_unobservable(f(predVertMap), predWidth, predHeight, f(obsVertMap), obsWidth, obsHeight, stream));
}

// ------------------------------------------------------------------------------------------------ intersection.h
int countSelfIntersections(const float4 * testSites,
                           const int nSites,
                           const SE3 * T_mfs,
                           const SE3 * T_fms,
                           const int * sdfFrames,
                           const Grid3D<float> * sdfs,
                           const int nSdfs,
                           const int * potentialIntersection) {
    int n = 0;
    DARTK_CALL(dartk_count_self_intersections_tables(f(testSites), nSites, T_mfs, T_fms, sdfFrames, sdfs, nSdfs, potentialIntersection, &n));
    return n;
}

void normEqnsSelfIntersection(const float4 * testSites,
                              const int nSites,
                              const int dims,
                              const MirroredModel & model,
                              const int * potentialIntersection,
                              float * result,
                              float * debugError) {
    DARTK_CALL(dartk_ne_self_intersection(f(testSites), nSites, dims, model.dartk(), potentialIntersection, result, 0));
    if (debugError)
        DARTK_CALL(dartk_debug_self_intersection_error(f(testSites), nSites, model.dartk(), potentialIntersection, debugError, 0));
}

void normEqnsSelfIntersectionReduced(const float4 * testSites,
                                     const int nSites,
                                     const int fullDims,
                                     const int redDims,
                                     const MirroredModel & model,
                                     const float * dtheta_dalpha,
                                     const int * potentialIntersection,
                                     float * result) {
    DARTK_CALL(dartk_ne_self_intersection_reduced(f(testSites), nSites, fullDims, redDims, model.dartk(), dtheta_dalpha,
                                                  potentialIntersection, result, 0));
}

void normEqnsSelfIntersectionParamMap(const float4 * testSites,
                                      const int nSites,
                                      const int fullDims,
                                      const int redDims,
                                      const MirroredModel & model,
                                      const int * dMapping,
                                      const int * potentialIntersection,
                                      float * result) {
    DARTK_CALL(dartk_ne_self_intersection_parammap(f(testSites), nSites, fullDims, redDims, model.dartk(), dMapping, potentialIntersection,
                                                   result, 0));
}

void normEqnsIntersection(const float4 * testSites,
                          const int nSites,
                          const int dims,
                          const SE3 T_ds,
                          const SE3 T_sd,
                          const MirroredModel & srcModel,
                          const MirroredModel & dstModel,
                          float * result,
                          float * debugError) {
    DARTK_CALL(dartk_ne_intersection(f(testSites), nSites, dims, f(T_ds), f(T_sd), srcModel.dartk(), dstModel.dartk(), result, 0));
    if (debugError)
        DARTK_CALL(dartk_debug_intersection_error(f(testSites), nSites, f(T_ds), srcModel.dartk(), dstModel.dartk(), debugError, 0));
}

void normEqnsIntersectionReduced(const float4 * testSites,
                                 const int nSites,
                                 const int fullDims,
                                 const int redDims,
                                 const SE3 T_ds,
                                 const SE3 T_sd,
                                 const MirroredModel & srcModel,
                                 const MirroredModel & dstModel,
                                 const float * dtheta_dalpha_src,
                                 float * result) {
    DARTK_CALL(dartk_ne_intersection_reduced(f(testSites), nSites, fullDims, redDims, f(T_ds), f(T_sd), srcModel.dartk(), dstModel.dartk(),
                                             dtheta_dalpha_src, result, 0));
}

void normEqnsIntersectionParamMap(const float4 * testSites,
                                  const int nSites,
                                  const int fullDims,
                                  const int redDims,
                                  const SE3 T_ds,
                                  const SE3 T_sd,
                                  const MirroredModel & srcModel,
                                  const MirroredModel & dstModel,
                                  const int * dMapping_src,
                                  float * result) {
    DARTK_CALL(dartk_ne_intersection_parammap(f(testSites), nSites, fullDims, redDims, f(T_ds), f(T_sd), srcModel.dartk(), dstModel.dartk(),
                                              dMapping_src, result, 0));
}

void initDebugIntersectionError(float * debugError,
                                const int nSites) {
    DARTK_CALL(dartk_init_debug_intersection_error(debugError, nSites, 0));
}

// ------------------------------------------------------------------------------------------------ raycast.h
void raycastPrediction(float2 fl,
                       float2 pp,
                       const int width,
                       const int height,
                       const int modelNum,
                       const SE3 T_mc,
                       const SE3 * T_fms,
                       const SE3 * T_mfs,
                       const int * sdfFrames,
                       const Grid3D<float> * sdfs,
                       const int nSdfs,
                       float4 * prediction,
                       const float levelSet,
                       hipStream_t stream) {
    DARTK_CALL(dartk_raycast_prediction_tables(fl.x, fl.y, pp.x, pp.y, width, height, modelNum, f(T_mc), T_fms, T_mfs, sdfFrames, sdfs, nSdfs,
                                               f(prediction), levelSet, stream));
}

// ------------------------------------------------------------------------------------------------ geometry/distance_transforms.h
// The two instantiations the path uses: distanceTransform3D<float,true> on the observation SDF (optimizer.cpp:210-213) and
// signedDistanceTransform3D<float,true> at model load (host_only_model.cpp:298-312). The scratch arguments are not needed.
template <typename Real, bool takeSqrt>
void distanceTransform3D(Real * in, Real * out, const unsigned int width, const unsigned int height, const unsigned int depth, Real * zScratch, int * vScratch);
template <>
void distanceTransform3D<float, true>(float * in, float * out, const unsigned int width, const unsigned int height, const unsigned int depth, float * zScratch, int * vScratch) {
    (void)zScratch; (void)vScratch;
    DARTK_CALL(dartk_distance_transform_3d(in, out, (int)width, (int)height, (int)depth, 1, 0));
}
template <typename Real, bool takeSqrt>
void signedDistanceTransform3D(Real * in, Real * out, const unsigned int width, const unsigned int height, const unsigned int depth);
template <>
void signedDistanceTransform3D<float, true>(float * in, float * out, const unsigned int width, const unsigned int height, const unsigned int depth) {
    DARTK_CALL(dartk_signed_distance_transform_3d(in, out, (int)width, (int)height, (int)depth, 1, 0));
}

// ------------------------------------------------------------------------------------------------ img_proc/organized_point_cloud.h
template <typename DepthType>
void depthToVertices(const DepthType * depthIn, float4 * vertOut, const int width, const int height, const float2 pp, const float2 fl, const float2 range);
template <>
void depthToVertices<float>(const float * depthIn, float4 * vertOut, const int width, const int height, const float2 pp, const float2 fl, const float2 range) {
    DARTK_CALL(dartk_depth_to_vertices(depthIn, f(vertOut), width, height, pp.x, pp.y, fl.x, fl.y, range.x, range.y, 0));
}

void verticesToNormals(const float4 * vertIn, float4 * normOut, const int width, const int height) {
    DARTK_CALL(dartk_vertices_to_normals(f(vertIn), f(normOut), width, height, 0));
}

}  // namespace dart
