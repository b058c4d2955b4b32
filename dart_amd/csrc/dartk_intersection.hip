// dartk_intersection.hip -- self- and inter-model intersection terms.
// Replaces: optimization/kernels/intersection.cu (kernels :11-597, host wrappers :715-947).
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_ne_bodies.h"

namespace dartk {

// SELF: articulation-only system of size dims-6 (intersection.cu:66-300).
template <int PPT, int RED>
__global__ void __launch_bounds__(NE_BLOCK)
ne_self_intersection_kernel(const float4 *__restrict__ sites, int nSites, ModelView m_in, const int *__restrict__ pot,
                            int redDims, const float *__restrict__ dtheta_dalpha, const int *__restrict__ mapping,
                            float *__restrict__ slabs) {
    ne_self_intersection_body<PPT, RED>(sites, nSites, m_in, pot, redDims, dtheta_dalpha, mapping, slabs, blockIdx.x, gridDim.x, blockIdx.y);
}

// INTER-MODEL: source cloud in destination SDFs, full source-pose system (intersection.cu:367-597).
template <int PPT, int RED>
__global__ void __launch_bounds__(NE_BLOCK)
ne_intersection_kernel(const float4 *__restrict__ sites, int nSites, ModelView src_in, ModelView dst_in,
                       const float *__restrict__ T_ds_sd /* [24] */, int redDims, const float *__restrict__ dtheta_dalpha,
                       const int *__restrict__ mapping, float *__restrict__ slabs, int tabStride) {
    ne_intersection_body<PPT, RED>(sites, nSites, src_in, dst_in, T_ds_sd, redDims, dtheta_dalpha, mapping, slabs, tabStride, blockIdx.x, gridDim.x, blockIdx.y);
}

// counters (intersection.cu:11-64, :315-365)
__global__ void count_self_kernel(const float4 *__restrict__ sites, int nSites, ModelView m, const int *__restrict__ pot, int *__restrict__ n) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (index < nSites) {
        const float4 v = sites[index];
        const int srcGrid = (int)roundf(v.w);
        const f3 vm = se3_apply_w(m.T_mf + 12 * m.sdfFrame[srcGrid], v.x, v.y, v.z, 1.0f);
        hit = first_collision(m, m.T_fm, vm, pot, srcGrid).hit;
    }
    const unsigned long long mask = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(n, __popcll(mask));
}
__global__ void count_inter_kernel(const float4 *__restrict__ sites, int nSites, ModelView src, ModelView dst,
                                   const float *__restrict__ T_ds, int *__restrict__ n) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (index < nSites) {
        const float4 v = sites[index];
        const int srcGrid = (int)roundf(v.w);
        const f3 vsm = se3_apply_w(src.T_mf + 12 * src.sdfFrame[srcGrid], v.x, v.y, v.z, 1.0f);
        const f3 vdm = se3_apply_w(T_ds, vsm.x, vsm.y, vsm.z, 1.0f);
        hit = first_collision(dst, dst.T_fm, vdm, nullptr, 0).hit;
    }
    const unsigned long long mask = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(n, __popcll(mask));
}

// the terms' optional debugError output (intersection.cu:131-133 self: = residual^2; :433-435 pairwise: += residual^2; sites
// without a collision are left as they are -- NaN after initDebugIntersectionError, :303-313)
__global__ void debug_error_self_kernel(const float4 *__restrict__ sites, int nSites, ModelView m, const int *__restrict__ pot,
                                        float *__restrict__ debugError) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= nSites) return;
    const float4 v = sites[index];
    const int srcGrid = (int)roundf(v.w);
    const f3 vm = se3_apply_w(m.T_mf + 12 * m.sdfFrame[srcGrid], v.x, v.y, v.z, 1.0f);
    const Collision c = first_collision(m, m.T_fm, vm, pot, srcGrid);
    if (c.hit) debugError[index] = c.residual * c.residual;
}
__global__ void debug_error_inter_kernel(const float4 *__restrict__ sites, int nSites, ModelView src, ModelView dst,
                                         const float *__restrict__ T_ds, float *__restrict__ debugError) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= nSites) return;
    const float4 v = sites[index];
    const int srcGrid = (int)roundf(v.w);
    const f3 vsm = se3_apply_w(src.T_mf + 12 * src.sdfFrame[srcGrid], v.x, v.y, v.z, 1.0f);
    const f3 vdm = se3_apply_w(T_ds, vsm.x, vsm.y, vsm.z, 1.0f);
    const Collision c = first_collision(dst, dst.T_fm, vdm, nullptr, 0);
    if (c.hit) debugError[index] += c.residual * c.residual;
}
__global__ void debug_error_init_kernel(float *__restrict__ debugError, int nSites) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index < nSites) debugError[index] = __int_as_float(0x7fc00000);
}

static int check_sites(const Model *m, int nSites) {
    DARTK_REQUIRE(m != nullptr, "null model");
    DARTK_REQUIRE(nSites >= 0, "negative site count");
    return DARTK_OK;
}

template <int RED>
static int launch_self(const float *d_sites, int nSites, int fullDims, int redDims, Model *mod, const float *d_A,
                       const int *d_map, const int *d_pot, float *d_result, hipStream_t st) {
    const ModelView &m = mod->v;
    DARTK_REQUIRE(fullDims == 6 + m.nJoints, "dims %d does not match the model's 6+%d", fullDims, m.nJoints);
    const int d = (RED ? redDims : fullDims) - 6;
    DARTK_REQUIRE(d >= 1 && d + 6 <= kMaxDims, "articulated system dimension %d out of range", d);
    const int S = sys_size(d);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const FrameBatch fb = cur_batch();
    if (nSites == 0) {
        for (int f = 0; f < fb.n; ++f) DARTK_HIP(hipMemsetAsync(d_result + (size_t)f * fb.sys, 0, S * sizeof(float), st));
        return DARTK_OK;
    }
    const int nTiles = (nSites + NE_BLOCK - 1) / NE_BLOCK;
    const int cap = fb.n > 1 ? 64 : kMaxPartialBlocks;      // a batch fills the chip with its frames
    const int nBlocks = nTiles < cap ? nTiles : cap;
    rc = ensure_partials(ws, (size_t)fb.n * nBlocks * S);
    if (rc) return rc;
    size_t words = 24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8;
    if (RED) words += (size_t)NE_BLOCK * ((fullDims - 6) | 1);
    const size_t ldsBytes = words * sizeof(float);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
#define DARTK_LAUNCH_SELF(PPT_)                                                                                 \
    do {                                                                                                        \
        auto kfn = ne_self_intersection_kernel<PPT_, RED>;                                                      \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks, fb.n), dim3(NE_BLOCK), ldsBytes, st, (const float4 *)d_sites, nSites, m, d_pot, \
                           redDims, d_A, d_map, lane_partials(ws));                                                  \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_SELF(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_SELF(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_SELF(6);
    else DARTK_LAUNCH_SELF(20);
#undef DARTK_LAUNCH_SELF
    DARTK_HIP(hipGetLastError());
    return launch_reduce_slabs(lane_partials(ws), nBlocks, S, d_result, 1, st);
}

// d_T_ds_sd: [T_ds | T_sd] (24 floats) on the device
template <int RED>
static int launch_inter_dev(const float *d_sites, int nSites, int fullDims, int redDims, const float *d_T_ds_sd,
                            Model *src, Model *dst, const float *d_A, const int *d_map, float *d_result, hipStream_t st);

template <int RED>
static int launch_inter(const float *d_sites, int nSites, int fullDims, int redDims, const float *h_T_ds,
                        const float *h_T_sd, Model *src, Model *dst, const float *d_A, const int *d_map,
                        float *d_result, hipStream_t st) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    float hT[24];
    for (int i = 0; i < 12; ++i) { hT[i] = h_T_ds[i]; hT[12 + i] = h_T_sd[i]; }
    DARTK_HIP(hipMemcpyAsync(ws->sysScratch + 64, hT, sizeof(hT), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipStreamSynchronize(st));
    return launch_inter_dev<RED>(d_sites, nSites, fullDims, redDims, ws->sysScratch + 64, src, dst, d_A, d_map, d_result, st);
}

template <int RED>
static int launch_inter_dev(const float *d_sites, int nSites, int fullDims, int redDims, const float *d_T_ds_sd,
                            Model *src, Model *dst, const float *d_A, const int *d_map, float *d_result, hipStream_t st) {
    const ModelView &ms = src->v;
    const ModelView &md = dst->v;
    DARTK_REQUIRE(fullDims == 6 + ms.nJoints, "dims %d does not match the source model's 6+%d", fullDims, ms.nJoints);
    const int d = RED ? redDims : fullDims;
    DARTK_REQUIRE(d >= 6 && d <= kMaxDims, "system dimension %d out of range", d);
    const int S = sys_size(d);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const FrameBatch fb = cur_batch();
    if (nSites == 0) {
        for (int f = 0; f < fb.n; ++f) DARTK_HIP(hipMemsetAsync(d_result + (size_t)f * fb.sys, 0, S * sizeof(float), st));
        return DARTK_OK;
    }
    const int nTiles = (nSites + NE_BLOCK - 1) / NE_BLOCK;
    const int cap = fb.n > 1 ? 64 : kMaxPartialBlocks;      // a batch fills the chip with its frames
    const int nBlocks = nTiles < cap ? nTiles : cap;
    rc = ensure_partials(ws, (size_t)fb.n * nBlocks * S);
    if (rc) return rc;
    size_t words = 24 * (size_t)(ms.nFrames + md.nFrames) + (size_t)NE_BLOCK * row_len(d, false) + 8;
    if (RED) words += (size_t)NE_BLOCK * (fullDims | 1);
    const size_t ldsBytes = words * sizeof(float);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
#define DARTK_LAUNCH_INTER(PPT_)                                                                                \
    do {                                                                                                        \
        auto kfn = ne_intersection_kernel<PPT_, RED>;                                                           \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks, fb.n), dim3(NE_BLOCK), ldsBytes, st, (const float4 *)d_sites, nSites, ms, md, \
                           d_T_ds_sd, redDims, d_A, d_map, lane_partials(ws), fb.tab);                               \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_INTER(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_INTER(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_INTER(6);
    else DARTK_LAUNCH_INTER(20);
#undef DARTK_LAUNCH_INTER
    DARTK_HIP(hipGetLastError());
    return launch_reduce_slabs(lane_partials(ws), nBlocks, S, d_result, 1, st);
}

size_t ne_self_lds_bytes(const ModelView &m, int redDims, int red) {
    const int fullDims = 6 + m.nJoints;
    const int d = (red ? redDims : fullDims) - 6;
    size_t words = 24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8;
    if (red) words += (size_t)NE_BLOCK * ((fullDims - 6) | 1);
    return words * sizeof(float);
}
size_t ne_inter_lds_bytes(const ModelView &src, const ModelView &dst, int redDims, int red) {
    const int fullDims = 6 + src.nJoints;
    const int d = red ? redDims : fullDims;
    size_t words = 24 * (size_t)(src.nFrames + dst.nFrames) + (size_t)NE_BLOCK * row_len(d, false) + 8;
    if (red) words += (size_t)NE_BLOCK * (fullDims | 1);
    return words * sizeof(float);
}

// T_ds = T_mc(dst) * T_cm(src) and its inverse from the models' device poses (optimizer.cpp:396-397)
__global__ void pair_transform_kernel(ModelView src, ModelView dst, float *__restrict__ out24, int tabStride) {
    if (threadIdx.x != 0) return;
    select_frame(src, blockIdx.x);                  // frame of a batch (0 otherwise)
    select_frame(dst, blockIdx.x);
    out24 += (size_t)blockIdx.x * tabStride;
    float A[12], B[12], T_ds[12], T_sd[12];
    for (int i = 0; i < 12; ++i) { A[i] = dst.T_mc[i]; B[i] = src.T_cm[i]; }
    se3_mul(A, B, T_ds);
    se3_invert(T_ds, T_sd);
    for (int i = 0; i < 12; ++i) { out24[i] = T_ds[i]; out24[12 + i] = T_sd[i]; }
}

// whole terms (kernel + reduction into d_result) for the device-resident multi-model loop; red: 0 full, 1 linear, 2 map
int gn_ne_self_any(int red, const float *d_sites, int nSites, int fullDims, int redDims, Model *mod, const float *d_A,
                   const int *d_map, const int *d_pot, float *d_result, hipStream_t st) {
    if (red == 1) return launch_self<1>(d_sites, nSites, fullDims, redDims, mod, d_A, nullptr, d_pot, d_result, st);
    if (red == 2) return launch_self<2>(d_sites, nSites, fullDims, redDims, mod, nullptr, d_map, d_pot, d_result, st);
    return launch_self<0>(d_sites, nSites, fullDims, fullDims, mod, nullptr, nullptr, d_pot, d_result, st);
}
int gn_ne_inter_any(int red, const float *d_sites, int nSites, int fullDims, int redDims, Model *src, Model *dst,
                    const float *d_A, const int *d_map, float *d_T_scratch24, float *d_result, hipStream_t st) {
    hipLaunchKernelGGL(pair_transform_kernel, dim3(cur_batch().n), dim3(64), 0, st, src->v, dst->v, d_T_scratch24, cur_batch().tab);
    if (red == 1) return launch_inter_dev<1>(d_sites, nSites, fullDims, redDims, d_T_scratch24, src, dst, d_A, nullptr, d_result, st);
    if (red == 2) return launch_inter_dev<2>(d_sites, nSites, fullDims, redDims, d_T_scratch24, src, dst, nullptr, d_map, d_result, st);
    return launch_inter_dev<0>(d_sites, nSites, fullDims, fullDims, d_T_scratch24, src, dst, nullptr, nullptr, d_result, st);
}

// entry point for the device-resident Gauss-Newton loops: kernel only, slabs chosen by the caller
int gn_launch_ne_self(const float *d_sites, int nSites, const ModelView &m, const int *d_pot, float *slabs, int nBlocks,
                      hipStream_t st) {
    const int fullDims = 6 + m.nJoints;
    const int d = fullDims - 6;
    const int S = sys_size(d);
    size_t words = 24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8;
    const size_t ldsBytes = words * sizeof(float);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
#define DARTK_LAUNCH_SELF(PPT_)                                                                                 \
    do {                                                                                                        \
        auto kfn = ne_self_intersection_kernel<PPT_, 0>;                                                        \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks, cur_batch().n), dim3(NE_BLOCK), ldsBytes, st, (const float4 *)d_sites, nSites, m, d_pot, \
                           fullDims, nullptr, nullptr, slabs);                                                  \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_SELF(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_SELF(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_SELF(6);
    else DARTK_LAUNCH_SELF(20);
#undef DARTK_LAUNCH_SELF
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

DARTK_COUNTER_ACCESS(intersection)

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_ne_self_intersection(const float *d_sites, int nSites, int dims, dartk_model_t h, const int32_t *d_pot,
                               float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE((d_sites || nSites == 0) && d_pot && d_result, "dartk_ne_self_intersection: null argument");
    int rc = check_sites(mod, nSites);
    if (rc) return rc;
    return launch_self<0>(d_sites, nSites, dims, dims, mod, nullptr, nullptr, d_pot, d_result, as_stream(stream));
}
int dartk_ne_self_intersection_reduced(const float *d_sites, int nSites, int fullDims, int redDims, dartk_model_t h,
                                       const float *d_A, const int32_t *d_pot, float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE((d_sites || nSites == 0) && d_pot && d_result && d_A, "dartk_ne_self_intersection_reduced: null argument");
    DARTK_REQUIRE(redDims > 6 && redDims <= fullDims, "redDims %d must be in (6,%d]", redDims, fullDims);
    int rc = check_sites(mod, nSites);
    if (rc) return rc;
    return launch_self<1>(d_sites, nSites, fullDims, redDims, mod, d_A, nullptr, d_pot, d_result, as_stream(stream));
}
int dartk_ne_self_intersection_parammap(const float *d_sites, int nSites, int fullDims, int redDims, dartk_model_t h,
                                        const int32_t *d_map, const int32_t *d_pot, float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE((d_sites || nSites == 0) && d_pot && d_result && d_map, "dartk_ne_self_intersection_parammap: null argument");
    DARTK_REQUIRE(redDims > 6 && redDims <= fullDims, "redDims %d must be in (6,%d]", redDims, fullDims);
    int rc = check_sites(mod, nSites);
    if (rc) return rc;
    return launch_self<2>(d_sites, nSites, fullDims, redDims, mod, nullptr, d_map, d_pot, d_result, as_stream(stream));
}

int dartk_ne_intersection(const float *d_sites, int nSites, int dims, const float *h_T_ds, const float *h_T_sd,
                          dartk_model_t hs, dartk_model_t hd, float *d_result, void *stream) {
    Model *src = reinterpret_cast<Model *>(hs), *dst = reinterpret_cast<Model *>(hd);
    DARTK_REQUIRE((d_sites || nSites == 0) && h_T_ds && h_T_sd && src && dst && d_result, "dartk_ne_intersection: null argument");
    return launch_inter<0>(d_sites, nSites, dims, dims, h_T_ds, h_T_sd, src, dst, nullptr, nullptr, d_result, as_stream(stream));
}
int dartk_ne_intersection_reduced(const float *d_sites, int nSites, int fullDims, int redDims, const float *h_T_ds,
                                  const float *h_T_sd, dartk_model_t hs, dartk_model_t hd, const float *d_A,
                                  float *d_result, void *stream) {
    Model *src = reinterpret_cast<Model *>(hs), *dst = reinterpret_cast<Model *>(hd);
    DARTK_REQUIRE((d_sites || nSites == 0) && h_T_ds && h_T_sd && src && dst && d_result && d_A, "dartk_ne_intersection_reduced: null argument");
    DARTK_REQUIRE(redDims >= 6 && redDims <= fullDims, "redDims %d must be in [6,%d]", redDims, fullDims);
    return launch_inter<1>(d_sites, nSites, fullDims, redDims, h_T_ds, h_T_sd, src, dst, d_A, nullptr, d_result, as_stream(stream));
}
int dartk_ne_intersection_parammap(const float *d_sites, int nSites, int fullDims, int redDims, const float *h_T_ds,
                                   const float *h_T_sd, dartk_model_t hs, dartk_model_t hd, const int32_t *d_map,
                                   float *d_result, void *stream) {
    Model *src = reinterpret_cast<Model *>(hs), *dst = reinterpret_cast<Model *>(hd);
    DARTK_REQUIRE((d_sites || nSites == 0) && h_T_ds && h_T_sd && src && dst && d_result && d_map, "dartk_ne_intersection_parammap: null argument");
    DARTK_REQUIRE(redDims >= 6 && redDims <= fullDims, "redDims %d must be in [6,%d]", redDims, fullDims);
    return launch_inter<2>(d_sites, nSites, fullDims, redDims, h_T_ds, h_T_sd, src, dst, nullptr, d_map, d_result, as_stream(stream));
}

int dartk_count_self_intersections(const float *d_sites, int nSites, dartk_model_t h, const int32_t *d_pot,
                                   int32_t *h_count, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE((d_sites || nSites == 0) && d_pot && h_count && mod, "dartk_count_self_intersections: null argument");
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    DARTK_HIP(hipMemsetAsync(ws->counters, 0, sizeof(int), st));
    if (nSites > 0)
        hipLaunchKernelGGL(count_self_kernel, dim3((nSites + 127) / 128), dim3(128), 0, st, (const float4 *)d_sites, nSites, mod->v, d_pot, ws->counters);
    DARTK_HIP(hipGetLastError());
    DARTK_HIP(hipMemcpyAsync(h_count, ws->counters, sizeof(int), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipStreamSynchronize(st));
    return DARTK_OK;
}

int dartk_init_debug_intersection_error(float *d_debugError, int nSites, void *stream) {
    DARTK_REQUIRE((d_debugError || nSites == 0) && nSites >= 0, "dartk_init_debug_intersection_error: bad argument");
    if (nSites > 0)
        hipLaunchKernelGGL(debug_error_init_kernel, dim3((nSites + 255) / 256), dim3(256), 0, as_stream(stream), d_debugError, nSites);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_debug_self_intersection_error(const float *d_sites, int nSites, dartk_model_t h, const int32_t *d_pot,
                                        float *d_debugError, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE((d_sites || nSites == 0) && d_pot && mod && (d_debugError || nSites == 0) && nSites >= 0,
                  "dartk_debug_self_intersection_error: bad argument");
    if (nSites > 0)
        hipLaunchKernelGGL(debug_error_self_kernel, dim3((nSites + 127) / 128), dim3(128), 0, as_stream(stream), (const float4 *)d_sites, nSites,
                           mod->v, d_pot, d_debugError);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_debug_intersection_error(const float *d_sites, int nSites, const float *h_T_ds, dartk_model_t hs, dartk_model_t hd,
                                   float *d_debugError, void *stream) {
    Model *src = reinterpret_cast<Model *>(hs), *dst = reinterpret_cast<Model *>(hd);
    DARTK_REQUIRE((d_sites || nSites == 0) && h_T_ds && src && dst && (d_debugError || nSites == 0) && nSites >= 0,
                  "dartk_debug_intersection_error: bad argument");
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    DARTK_HIP(hipMemcpyAsync(ws->sysScratch + 64, h_T_ds, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipStreamSynchronize(st));
    if (nSites > 0)
        hipLaunchKernelGGL(debug_error_inter_kernel, dim3((nSites + 127) / 128), dim3(128), 0, st, (const float4 *)d_sites, nSites, src->v, dst->v,
                           ws->sysScratch + 64, d_debugError);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_count_intersections(const float *d_sites, int nSites, const float *h_T_ds, dartk_model_t hs, dartk_model_t hd,
                              int32_t *h_count, void *stream) {
    Model *src = reinterpret_cast<Model *>(hs), *dst = reinterpret_cast<Model *>(hd);
    DARTK_REQUIRE((d_sites || nSites == 0) && h_T_ds && h_count && src && dst, "dartk_count_intersections: null argument");
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    DARTK_HIP(hipMemsetAsync(ws->counters, 0, sizeof(int), st));
    DARTK_HIP(hipMemcpyAsync(ws->sysScratch + 64, h_T_ds, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipStreamSynchronize(st));
    if (nSites > 0)
        hipLaunchKernelGGL(count_inter_kernel, dim3((nSites + 127) / 128), dim3(128), 0, st, (const float4 *)d_sites, nSites, src->v, dst->v, ws->sysScratch + 64, ws->counters);
    DARTK_HIP(hipGetLastError());
    DARTK_HIP(hipMemcpyAsync(h_count, ws->counters, sizeof(int), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipStreamSynchronize(st));
    return DARTK_OK;
}

}  // extern "C"
