// stub_kat1.cpp -- the reference's kernel test (test/test_obs_to_mod_kernels.cpp:14-109) written the way the reference writes
// it -- `dart::errorAndDataAssociation(dObsVertMap, dObsNormMap, width, height, model, opts, dPts, dLastElement, hLastElement,
// debugDataAssociation, debugError)` on a MirroredModel and OptimizationOptions -- and linked against integration/kernels_hip.cpp
// (the forwarding stubs with the reference's parameter lists) + libdartk.so. Plain g++: no HIP headers, no torch.
// Then a second pass over the same map through the other wrappers of the single-model path the way Optimizer::optimizePose calls
// them (optimizer.cpp:447-480): splatObsSdfZeros + distanceTransform3D<float,true>, raycastPrediction, cullUnobservable_,
// normEqnsObsToMod, normEqnsModToObs, errorModToObs -- checked for consistency with the direct C-ABI calls.
// Exit code 0 = all checks passed; 77 = no GPU.
//   g++ -O1 -std=c++11 -I include -I integration tests/cpp/stub_kat1.cpp integration/kernels_hip.cpp -L dart_amd -ldartk
//       -Wl,-rpath,'$ORIGIN/../../dart_amd' -o tests/cpp/stub_kat1
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "dart_compat.h"

// the declarations of the reference's headers (obsToMod.h:12-21,75-86; modToObs.h:13-28,68-74,84-90; raycast.h:9-22;
// distance_transforms.h) with their default arguments
namespace dart {
void normEqnsObsToMod(const int dims, const float4 *dObsVertMap, const int width, const int height, const MirroredModel &model,
                      const OptimizationOptions &opts, DataAssociatedPoint *dPts, int nElements, float *dResult, float4 *debugJs = 0);
void errorAndDataAssociation(const float4 *dObsVertMap, const float4 *dObsNormMap, const int width, const int height,
                             const MirroredModel &model, const OptimizationOptions &opts, DataAssociatedPoint *dPts, int *dLastElement,
                             int *hLastElement, int *debugDataAssociation = 0, float *debugError = 0, float4 *debugNorm = 0);
float errorModToObs(const float4 *dLabeledPredictedVertMap, const int width, const int height, const Grid3D<float> *dObsSdf);
void normEqnsModToObs(const int dimensions, const float4 *dLabeledPredictedVertMap, const int width, const int height,
                      const MirroredModel &model, const SE3 T_gc, float *dResult, int *numPredictions = 0, int *debugDataAssociation = 0,
                      float *debugError = 0, float4 *debugNorm = 0);
void splatObsSdfZeros(const float4 *dObsVertMap, const int width, const int height, const SE3 &T_cm, const Grid3D<float> *dObsSdf,
                      const uint3 sdfDim, const float focalLength);
void cullUnobservable_(float4 *predVertMap, const int predWidth, const int predHeight, const float4 *obsVertMap, const int obsWidth,
                       const int obsHeight, const hipStream_t stream = 0);
void raycastPrediction(float2 fl, float2 pp, const int width, const int height, const int modelNum, const SE3 T_mc, const SE3 *T_fms,
                       const SE3 *T_mfs, const int *sdfFrames, const Grid3D<float> *sdfs, const int nSdfs, float4 *prediction,
                       const float levelSet, hipStream_t stream);
template <typename Real, bool takeSqrt>
void distanceTransform3D(Real *in, Real *out, const unsigned int width, const unsigned int height, const unsigned int depth, Real *zScratch,
                         int *vScratch);
template <>
void distanceTransform3D<float, true>(float *in, float *out, const unsigned int width, const unsigned int height, const unsigned int depth,
                                      float *zScratch, int *vScratch);
}  // namespace dart

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dartk_last_error()); return rc_ == DARTK_ERR_NO_DEVICE ? 77 : 1; } \
    } while (0)

template <typename T>
static T *deviceArray(size_t n) {
    void *p = 0;
    if (dartk_buf_alloc(&p, n * sizeof(T)) != 0) { fprintf(stderr, "alloc failed: %s\n", dartk_last_error()); exit(1); }
    return static_cast<T *>(p);
}

int main() {
    CHECK(dartk_init(0));
    const float sphereRadius = 0.1f, sdfResolution = 0.005f, padding = 0.1f;
    const int dim = (int)std::ceil((2 * sphereRadius + 2 * padding) / sdfResolution);
    const float offset = -(sphereRadius + padding);
    std::vector<float> sdf((size_t)dim * dim * dim);
    for (int z = 0; z < dim; ++z) for (int y = 0; y < dim; ++y) for (int x = 0; x < dim; ++x) {
        const float cx = offset + sdfResolution * (x + 0.5f), cy = offset + sdfResolution * (y + 0.5f), cz = offset + sdfResolution * (z + 0.5f);
        sdf[x + (size_t)dim * (y + (size_t)dim * z)] = (std::sqrt(cx * cx + cy * cy + cz * cz) - sphereRadius) / sdfResolution;
    }
    dartk_model_desc desc;
    memset(&desc, 0, sizeof(desc));
    const int32_t parent[1] = {-1}, sdfFrame[1] = {0}, sdfDim[3] = {dim, dim, dim}, deps[1] = {0};
    const float sdfOff[3] = {offset, offset, offset}, sdfRes[1] = {sdfResolution};
    const float *sdfData[1] = {sdf.data()};
    desc.n_frames = 1; desc.n_joints = 0; desc.n_sdfs = 1; desc.model_id = 0;
    desc.frame_parent = parent; desc.sdf_frame = sdfFrame; desc.dependencies = deps;
    desc.sdf_dim = sdfDim; desc.sdf_offset = sdfOff; desc.sdf_resolution = sdfRes; desc.sdf_data = sdfData;
    // an observation SDF for the second pass: 32^3 voxels of 1.25 cm around the sphere (model frame)
    const int od = 32;
    desc.obs_sdf_dim[0] = desc.obs_sdf_dim[1] = desc.obs_sdf_dim[2] = od;
    desc.obs_sdf_offset[0] = desc.obs_sdf_offset[1] = desc.obs_sdf_offset[2] = -0.2f;
    desc.obs_sdf_resolution = 0.0125f;
    dartk_model_t handle;
    CHECK(dartk_model_create(&desc, &handle));
    dart::MirroredModel model(handle, 0, 1);         // (the reference's constructor builds the handle from a HostOnlyModel)
    if (dartk_model_dims(handle) != 6) { fprintf(stderr, "pose dimensionality %d != 6\n", dartk_model_dims(handle)); return 1; }

    // ---------------------------------------------------------------- test_obs_to_mod_kernels.cpp:42-102
    const int obsWidth = 10, obsHeight = 10, n = obsWidth * obsHeight;
    std::vector<float4> obsVertMap(n), obsNormMap(n, make_float4(0, 0, 0, 0));
    std::default_random_engine generator;
    std::bernoulli_distribution validDistribution(0.8);
    std::normal_distribution<float> normalDistribution(0, 0.1);
    for (int i = 0; i < n; ++i) {
        if (validDistribution(generator))
            obsVertMap[i] = make_float4(normalDistribution(generator), normalDistribution(generator), normalDistribution(generator), 1);
        else
            obsVertMap[i] = make_float4(0, 0, 0, 0);
    }
    float4 *dObsVertMap = deviceArray<float4>(n), *dObsNormMap = deviceArray<float4>(n);
    dart::DataAssociatedPoint *dPts = deviceArray<dart::DataAssociatedPoint>(n);
    int *dLastElement = deviceArray<int>(8), *dDebugDA = deviceArray<int>(n);
    float *dDebugError = deviceArray<float>(n);
    CHECK(dartk_buf_h2d(dObsVertMap, obsVertMap.data(), n * sizeof(float4), NULL));
    CHECK(dartk_buf_h2d(dObsNormMap, obsNormMap.data(), n * sizeof(float4), NULL));
    const dart::SE3 identity;
    CHECK(dartk_model_set_pose(handle, reinterpret_cast<const float *>(&identity), NULL, NULL));
    dart::OptimizationOptions opts;
    opts.distThreshold[0] = 0.05f;
    int lastElement = -1;
    dart::errorAndDataAssociation(dObsVertMap, dObsNormMap, obsWidth, obsHeight, model, opts, dPts, dLastElement, &lastElement, dDebugDA,
                                  dDebugError);
    std::vector<int> debugDataAssociation(n);
    std::vector<float> debugError(n);
    CHECK(dartk_buf_d2h(debugDataAssociation.data(), dDebugDA, n * sizeof(int), NULL));
    CHECK(dartk_buf_d2h(debugError.data(), dDebugError, n * sizeof(float), NULL));
    int failures = 0, associated = 0;
    for (int i = 0; i < n; ++i) {
        const float4 &p = obsVertMap[i];
        if (p.w == 0) {
            if (debugDataAssociation[i] != -1 || !std::isnan(debugError[i])) { ++failures; fprintf(stderr, "pixel %d: invalid point associated\n", i); }
        } else {
            const float dist = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z) - sphereRadius;
            if (std::fabs(dist) > opts.distThreshold[0]) {
                if (debugDataAssociation[i] != -1 || !std::isnan(debugError[i])) { ++failures; fprintf(stderr, "pixel %d: far point associated\n", i); }
            } else {
                ++associated;
                if (debugDataAssociation[i] != 0) { ++failures; fprintf(stderr, "pixel %d: DA %d != 0\n", i, debugDataAssociation[i]); }
                if (!(std::fabs(dist - debugError[i]) <= sdfResolution / 8)) { ++failures; fprintf(stderr, "pixel %d: error %g vs %g\n", i, debugError[i], dist); }
            }
        }
    }
    if (lastElement != associated) { ++failures; fprintf(stderr, "count %d != %d\n", lastElement, associated); }

    // ---------------------------------------------------------------- the other wrappers of the single-model path
    // normEqnsObsToMod == dartk_ne_obs2mod on the same list
    const int S = 6 + 21 + 1;
    float *dResult = deviceArray<float>(S), *dResult2 = deviceArray<float>(S);
    dart::normEqnsObsToMod(6, dObsVertMap, obsWidth, obsHeight, model, opts, dPts, lastElement, dResult);
    CHECK(dartk_ne_obs2mod(6, reinterpret_cast<const float *>(dObsVertMap), obsWidth, obsHeight, handle, opts.huberDelta,
                           reinterpret_cast<const dartk_da_point *>(dPts), lastElement, dResult2, NULL));
    std::vector<float> r1(S), r2(S);
    CHECK(dartk_buf_d2h(r1.data(), dResult, S * sizeof(float), NULL));
    CHECK(dartk_buf_d2h(r2.data(), dResult2, S * sizeof(float), NULL));
    if (memcmp(r1.data(), r2.data(), S * sizeof(float)) != 0 || !(r1[S - 1] > 0)) { ++failures; fprintf(stderr, "normEqnsObsToMod differs from dartk_ne_obs2mod\n"); }

    // the sphere 0.6 m in front of the camera: ray-cast it, use the result as the observation, build the observation SDF the way
    // generateObsSdfSplatZeroAndDistanceTransform does (optimizer.cpp:168-219), then the mod->obs term at the same pose
    dart::SE3 T_cm(make_float4(1, 0, 0, 0), make_float4(0, 1, 0, 0), make_float4(0, 0, 1, 0.6f));
    dart::SE3 T_mc(make_float4(1, 0, 0, 0), make_float4(0, 1, 0, 0), make_float4(0, 0, 1, -0.6f));
    CHECK(dartk_model_set_pose(handle, reinterpret_cast<const float *>(&T_mc), NULL, NULL));
    const int pw = 80, ph = 60, ow = 160, oh = 120;
    float4 *dPred = deviceArray<float4>(pw * ph), *dObs = deviceArray<float4>(ow * oh), *dObsN = deviceArray<float4>(ow * oh);
    float *dDepth = deviceArray<float>(ow * oh);
    dart::raycastPrediction(make_float2(268.5f, 268.5f), make_float2(80, 60), ow, oh, 0, T_mc, model.getDeviceTransformsModelToFrame(),
                            model.getDeviceTransformsFrameToModel(), model.getDeviceSdfFrames(), model.getDeviceSdfs(), model.getNumSdfs(),
                            dObs, 0.0f, 0);
    CHECK(dartk_prediction_to_depth(reinterpret_cast<const float *>(dObs), dDepth, ow, oh, NULL));
    CHECK(dartk_depth_to_vertices(dDepth, reinterpret_cast<float *>(dObs), ow, oh, 80, 60, 268.5f, 268.5f, 0.1f, 5.0f, NULL));
    CHECK(dartk_vertices_to_normals(reinterpret_cast<const float *>(dObs), reinterpret_cast<float *>(dObsN), ow, oh, NULL));
    dart::splatObsSdfZeros(dObs, ow, oh, T_cm, model.getDeviceObsSdf(), make_uint3(od, od, od), 268.5f);
    float *dTmp = deviceArray<float>((size_t)od * od * od);
    dart::distanceTransform3D<float, true>(model.getDeviceObsSdfData(), dTmp, od, od, od, 0, 0);
    std::vector<float> viaStubs((size_t)od * od * od), viaBuild((size_t)od * od * od);
    CHECK(dartk_buf_d2h(viaStubs.data(), dTmp, viaStubs.size() * sizeof(float), NULL));
    CHECK(dartk_obs_sdf_build(reinterpret_cast<const float *>(dObs), ow, oh, handle, 268.5f, NULL));
    CHECK(dartk_buf_d2h(viaBuild.data(), model.getDeviceObsSdfData(), viaBuild.size() * sizeof(float), NULL));
    if (memcmp(viaStubs.data(), viaBuild.data(), viaStubs.size() * sizeof(float)) != 0) { ++failures; fprintf(stderr, "splatObsSdfZeros + distanceTransform3D differs from dartk_obs_sdf_build\n"); }
    int zeros = 0;
    for (size_t i = 0; i < viaBuild.size(); ++i) zeros += viaBuild[i] == 0;
    if (zeros < 20) { ++failures; fprintf(stderr, "observation SDF has only %d zero voxels\n", zeros); }
    dart::raycastPrediction(make_float2(134.25f, 134.25f), make_float2(40, 30), pw, ph, 0, T_mc, model.getDeviceTransformsModelToFrame(),
                            model.getDeviceTransformsFrameToModel(), model.getDeviceSdfFrames(), model.getDeviceSdfs(), model.getNumSdfs(),
                            dPred, 0.0f, 0);
    dart::cullUnobservable_(dPred, pw, ph, dObs, ow, oh);
    int *dNum = deviceArray<int>(1);
    dart::normEqnsModToObs(6, dPred, pw, ph, model, T_mc, dResult, dNum);
    CHECK(dartk_ne_mod2obs(6, reinterpret_cast<const float *>(dPred), pw, ph, handle, reinterpret_cast<const float *>(&T_mc), dResult2, dNum, NULL));
    int numPredictions = 0;
    CHECK(dartk_buf_d2h(&numPredictions, dNum, sizeof(int), NULL));
    CHECK(dartk_buf_d2h(r1.data(), dResult, S * sizeof(float), NULL));
    CHECK(dartk_buf_d2h(r2.data(), dResult2, S * sizeof(float), NULL));
    if (memcmp(r1.data(), r2.data(), S * sizeof(float)) != 0) { ++failures; fprintf(stderr, "normEqnsModToObs differs from dartk_ne_mod2obs\n"); }
    if (numPredictions < 100) { ++failures; fprintf(stderr, "only %d predicted points\n", numPredictions); }
    const float e = dart::errorModToObs(dPred, pw, ph, model.getDeviceObsSdf());
    if (!(e >= 0) || !std::isfinite(e)) { ++failures; fprintf(stderr, "errorModToObs = %g\n", e); }

    CHECK(dartk_stream_sync(NULL));
    void *bufs[] = {dObsVertMap, dObsNormMap, dPts, dLastElement, dDebugDA, dDebugError, dResult, dResult2, dPred, dObs, dObsN, dDepth, dTmp, dNum};
    for (void *b : bufs) CHECK(dartk_buf_free(b));
    CHECK(dartk_model_destroy(handle));
    printf("stub_kat1: %d of %d valid points associated, %d predicted points, %d failed checks "
           "(reference call syntax through integration/kernels_hip.cpp + libdartk.so)\n", associated, 80, numPredictions, failures);
    return failures ? 1 : 0;
}
