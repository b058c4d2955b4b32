// dartk_imgproc.hip -- the two sensor-front-end steps that sit directly before the hot path, needed by the
// on-device synthetic-depth generator: depth -> vertex map and vertex map -> normal map.
// Replaces: img_proc/organized_point_cloud.cu:11-40 (gpu_depthToVertices<float>), :265-305 (gpu_verticesToNormals).
#include "dartk_internal.h"

namespace dartk {

__global__ void depth_to_vertices_kernel(const float *__restrict__ depth, float4 *__restrict__ vert, int width, int height,
                                         float ppx, float ppy, float flx, float fly, float rmin, float rmax) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= width * height) return;
    const int u = index % width, v = index / width;
    const float d = depth[index];
    if (d >= rmin && d <= rmax) vert[index] = make_float4(((float)u - ppx) * (d / flx), ((float)v - ppy) * (d / fly), d, 1.0f);
    else vert[index].w = 0;
}

__global__ void vertices_to_normals_kernel(const float4 *__restrict__ vertIn, float4 *__restrict__ normOut, int width, int height) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= width * height) return;
    const int x = index % width, y = index / width;
    const float4 v = vertIn[index];
    if (v.w == 0) { normOut[index] = make_float4(0, 0, 0, 0); return; }
    float4 vLeft = vertIn[x == 0 ? index : index - 1];
    float4 vRight = vertIn[x == width - 1 ? index : index + 1];
    float4 vUp = vertIn[y == 0 ? index : index - width];
    float4 vDown = vertIn[y == height - 1 ? index : index + width];
    if (vRight.w == 0) vRight = v;
    if (vLeft.w == 0) vLeft = v;
    if (vDown.w == 0) vDown = v;
    if (vUp.w == 0) vUp = v;
    f3 vX, vY;
    vX.x = vRight.x - vLeft.x; vX.y = vRight.y - vLeft.y; vX.z = vRight.z - vLeft.z;
    vY.x = vDown.x - vUp.x; vY.y = vDown.y - vUp.y; vY.z = vDown.z - vUp.z;
    const f3 n = cross(vY, vX);
    const float len2 = dot(n, n);
    if (len2 > 0) {
        const float invLen = 1.0f / sqrtf(len2);
        normOut[index] = make_float4(n.x * invLen, n.y * invLen, n.z * invLen, 1);
    } else {
        normOut[index] = make_float4(0, 0, 0, 0);
    }
}

// synthetic depth: z of a ray-cast prediction map (0 where no surface was hit)
__global__ void prediction_to_depth_kernel(const float4 *__restrict__ pred, float *__restrict__ depth, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) depth[i] = pred[i].z;
}

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_depth_to_vertices(const float *d_depth, float *d_vert, int width, int height, float ppx, float ppy, float flx,
                            float fly, float rangeMin, float rangeMax, void *stream) {
    DARTK_REQUIRE(d_depth && d_vert && width > 0 && height > 0 && flx != 0 && fly != 0, "dartk_depth_to_vertices: bad argument");
    const int n = width * height;
    hipLaunchKernelGGL(depth_to_vertices_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), d_depth, (float4 *)d_vert,
                       width, height, ppx, ppy, flx, fly, rangeMin, rangeMax);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_vertices_to_normals(const float *d_vert, float *d_norm, int width, int height, void *stream) {
    DARTK_REQUIRE(d_vert && d_norm && width > 0 && height > 0, "dartk_vertices_to_normals: bad argument");
    const int n = width * height;
    hipLaunchKernelGGL(vertices_to_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), (const float4 *)d_vert,
                       (float4 *)d_norm, width, height);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_prediction_to_depth(const float *d_pred, float *d_depth, int width, int height, void *stream) {
    DARTK_REQUIRE(d_pred && d_depth && width > 0 && height > 0, "dartk_prediction_to_depth: bad argument");
    const int n = width * height;
    hipLaunchKernelGGL(prediction_to_depth_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), (const float4 *)d_pred, d_depth, n);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

}  // extern "C"
