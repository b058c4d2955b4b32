// dart_compat.h -- the few declarations of the reference tree that its kernel headers' parameter lists name, so that
// integration/kernels_hip.cpp (the forwarding stubs a DART maintainer drops in place of obsToMod.cu / modToObs.cu /
// intersection.cu / raycast.cu / distance_transforms.cu / organized_point_cloud.cu) compiles in THIS repository with plain g++:
// no CUDA, no Eigen, no Pangolin (none of them exist in the image). In the reference tree the stubs include the tree's own
// headers instead (geometry/SE3.h, geometry/grid_3d.h, optimization/optimization.h, model/mirrored_model.h, util/dart_types.h)
// and only MirroredModel changes: it owns a dartk_model_t and its device getters hand out the library's arrays.
//
// What is declared here and where it comes from (paths relative to the reference tree):
//   float2 / float3 / float4 / uint3    CUDA/HIP vector types (vector_types.h); same member names and layout
//   SE3                                 geometry/SE3.h:17-23 (three float4 rows [R|t] = 12 floats row-major)
//   Grid3D<T>                           geometry/grid_3d.h:19,423-426 (only ever a pointer type in these parameter lists)
//   JointType                           util/dart_types.h:6-9
//   OptimizationOptions, DataAssociatedPoint   optimization/optimization.h:9-72, 85-89 (fields the wrappers read)
//   MirroredModel                       model/mirrored_model.h:11-125 (the members the wrappers and the trackers' tables use)
//   hipStream_t                         the HIP spelling of the cudaStream_t in the reference's parameter lists
#ifndef DART_COMPAT_H
#define DART_COMPAT_H

#include <vector>

#include "dartk.h"

typedef struct ihipStream_t *hipStream_t;      // as in <hip/hip_runtime_api.h>

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
typedef unsigned int uint;
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { uint3 r = {x, y, z}; return r; }

namespace dart {

enum JointType { RotationalJoint = 0, PrismaticJoint = 1 };      // util/dart_types.h:6-9

struct SE3 {                                                      // geometry/SE3.h:17-23
    float4 r0, r1, r2;
    SE3() : r0(make_float4(1, 0, 0, 0)), r1(make_float4(0, 1, 0, 0)), r2(make_float4(0, 0, 1, 0)) {}
    SE3(const float4 &a, const float4 &b, const float4 &c) : r0(a), r1(b), r2(c) {}
};

template <typename T>
struct Grid3D {                                                   // geometry/grid_3d.h:423-426
    uint3 dim;
    float3 offset;
    float resolution;
    T *data;
};

struct OptimizationOptions {                                      // optimization/optimization.h:9-72 (same defaults)
    int numIterations;
    std::vector<float> distThreshold, regularization;
    float contactRegularization, lambdaObsToMod, lambdaModToObs;
    std::vector<float> lambdaIntersection, planeOffset;
    std::vector<float3> planeNormal;
    std::vector<float> regularizationScaled;
    float focalLength, normThreshold, huberDelta, contactThreshold;
    bool debugObsToModDA, debugObsToModErr, debugModToObsDA, debugModToObsErr, debugObsToModNorm, debugModToObsNorm, debugObsToModJs,
         debugIntersectionErr, debugJTJ;
    OptimizationOptions()
        : numIterations(5), distThreshold(1, 0.03f), regularization(1, 1e-20f), contactRegularization(1e-20f), lambdaObsToMod(1.0f),
          lambdaModToObs(1.0f), lambdaIntersection(1, 0.0f), planeOffset(1, -0.03f), planeNormal(1, make_float3(0, 0, 0)),
          regularizationScaled(1, 1.0f), focalLength(537.0f), normThreshold(-0.1f), huberDelta(0.02f), contactThreshold(0.01f),
          debugObsToModDA(false), debugObsToModErr(false), debugModToObsDA(false), debugModToObsErr(false), debugObsToModNorm(false),
          debugModToObsNorm(false), debugObsToModJs(false), debugIntersectionErr(false), debugJTJ(false) {}
};

struct DataAssociatedPoint { int index; int dataAssociation; float error; };      // optimization/optimization.h:85-89

// model/mirrored_model.h: what is left of the device mirror when libdartk.so owns the device state. The constructor of the
// reference (mirrored_model.cpp:8-120) fills a dartk_model_desc from the HostOnlyModel and calls dartk_model_create
// (INTEGRATION.md section 1); here a test hands in the finished handle.
class MirroredModel {
public:
    MirroredModel(dartk_model_t handle, int modelID, int numSdfs) : _dartk(handle), _modelID((uint)modelID), _numSdfs((uint)numSdfs) {}
    dartk_model_t dartk() const { return _dartk; }
    uint getModelID() const { return _modelID; }                                  // mirrored_model.h:58
    uint getNumSdfs() const { return _numSdfs; }                                  // :31
    // device getters (:65-87): the library's own arrays, same element layouts (SE3 = 12 floats); the SDF table and the
    // observation SDF are opaque to host code here as Grid3D<float>* is in the reference (they only travel back into kernels)
    const SE3 *getDeviceTransformsFrameToModel() const { return (const SE3 *)dartk_model_device_T_mf(_dartk); }
    const SE3 *getDeviceTransformsModelToFrame() const { return (const SE3 *)dartk_model_device_T_fm(_dartk); }
    const int *getDeviceFrameParents() const { return dartk_model_device_frame_parents(_dartk); }
    const int *getDeviceSdfFrames() const { return dartk_model_device_sdf_frames(_dartk); }
    const Grid3D<float> *getDeviceSdfs() const { return (const Grid3D<float> *)dartk_model_device_sdfs(_dartk); }
    const JointType *getDeviceJointTypes() const { return (const JointType *)dartk_model_device_joint_types(_dartk); }
    const float3 *getDeviceJointAxes() const { return (const float3 *)dartk_model_device_joint_axes(_dartk); }
    const Grid3D<float> *getDeviceObsSdf() const { return (const Grid3D<float> *)dartk_model_obs_sdf_data(_dartk); }
    float *getDeviceObsSdfData() { return dartk_model_obs_sdf_data(_dartk); }

private:
    dartk_model_t _dartk;
    uint _modelID, _numSdfs;
};

}  // namespace dart

#endif
