// dartk_obs2mod.hip -- observation-to-model data association and its normal equations.
// Replaces: optimization/kernels/obsToMod.cu (kernels :12-501, host wrappers :504-817).
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_ne_bodies.h"

namespace dartk {

// -----------------------------------------------------------------------------------------------------
// association, single model (obsToMod.cu:12-114). One lane per pixel, 256-lane workgroups over the
// row-major map: each wave reads 64 consecutive float4 = 1 KiB per load instruction. The loop over SDFs is
// wave-uniform, so SDF headers and frame transforms come through the scalar cache. The append is one
// atomic per wave (ballot + popcount), not one per pixel.
// -----------------------------------------------------------------------------------------------------
template <bool DBG>
__global__ void __launch_bounds__(256)
assoc_single_kernel(const float4 *__restrict__ vert, const float4 *__restrict__ norm, int nPix, ModelView m,
                    float distThr, float normThr, float planeOffset, float pnx, float pny, float pnz,
                    int *__restrict__ lastElement, dartk_da_point *__restrict__ pts, int *__restrict__ dbgDA,
                    float *__restrict__ dbgErr, float4 *__restrict__ dbgNorm) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    bool emit = false;
    int grid = -1;
    float sdfError = 1e20f;
    if (index < nPix) {
        if (DBG) {
            if (dbgDA) dbgDA[index] = -1;
            if (dbgErr) dbgErr[index] = __builtin_nanf("");
            if (dbgNorm) dbgNorm[index] = make_float4(0, 0, 0, 0);
        }
        const float4 xc = vert[index];
        if (xc.w > 0) {
            const AssocResult r = associate_point(m, xc, distThr, planeOffset, pnx, pny, pnz);
            grid = r.grid; sdfError = r.err;
            if (r.grid >= 0) {
                // normal test (obsToMod.cu:78-96)
                float4 nPred;
                emit = normal_test(m, r, norm[index], normThr, nPred);
                if (DBG && dbgNorm) dbgNorm[index] = nPred;
                if (emit && DBG) {
                    if (dbgDA) dbgDA[index] = grid;
                    if (dbgErr) dbgErr[index] = sdfError;
                }
            }
        }
    }
    const unsigned long long mask = __ballot(emit);
    if (mask) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(lastElement, __popcll(mask));
        base = __shfl(base, 0);
        if (emit) {
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            dartk_da_point p;
            p.index = index; p.dataAssociation = grid; p.error = sdfError;
            pts[base + __popcll(mask & lt)] = p;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// association, multi model (obsToMod.cu:116-244): threshold inside the arg-min, no normal test.
// -----------------------------------------------------------------------------------------------------
// STAGED (the loops' launches: a chip-sized grid whose workgroups stride over the valid-pixel list): every model's transforms,
// SDF -> frame table and SDF headers are copied to LDS once per workgroup; the per-(pixel, SDF) loop then reads them there instead
// of through two dependent scalar loads per SDF (frame index, then its transform) and a header fetch from memory.
template <bool DBG, bool STAGED = false>
__global__ void __launch_bounds__(256)
assoc_multi_kernel(const float4 *__restrict__ vert, int nPix, const MultiArgs *__restrict__ args,
                   int *__restrict__ lastElement, int *__restrict__ dbgDA, float *__restrict__ dbgErr, FrameBatch fb,
                   const int *__restrict__ validIdx, const int *__restrict__ nValid) {
    assoc_multi_body<DBG, STAGED>(vert, nPix, args, lastElement, dbgDA, dbgErr, fb, validIdx, nValid, blockIdx.x, gridDim.x, blockIdx.y);
}

// -----------------------------------------------------------------------------------------------------
// normal equations over the associated-point list (obsToMod.cu:246-501)
// RED: 0 full pose, 1 linear reduction, 2 parameter map.
// LDS: [T_fm | T_mf] (24*nFrames) | SdfDev table (16*nSdfs words) | scratch J (NE_BLOCK*fullDims, reduced only)
//      | rows (NE_BLOCK*ld) | wave counters
// -----------------------------------------------------------------------------------------------------
template <int PPT, int RED>
__global__ void __launch_bounds__(NE_BLOCK)
ne_obs2mod_kernel(const dartk_da_point *__restrict__ pts, const float4 *__restrict__ vert, int nPoints, ModelView m_in,
                  float huberDelta, int redDims, const float *__restrict__ dtheta_dalpha,
                  const int *__restrict__ mapping, float *__restrict__ slabs, const int *__restrict__ d_count, FrameBatch fb) {
    ne_obs2mod_body<PPT, RED>(pts, vert, nPoints, m_in, huberDelta, redDims, dtheta_dalpha, mapping, slabs, d_count, fb, blockIdx.x, gridDim.x, blockIdx.y);
}

size_t ne_obs2mod_lds_bytes(const ModelView &m, int d, int red) {
    const int fullDims = 6 + m.nJoints;
    size_t words = 24 * (size_t)m.nFrames + (sizeof(SdfDev) / 4) * (size_t)m.nSdfs + (size_t)NE_BLOCK * row_len(d, red != 0);
    if (red) words += (size_t)NE_BLOCK * (fullDims | 1);
    return words * sizeof(float);
}

template <int RED>
static int launch_ne_obs2mod(int fullDims, int redDims, const float *d_vert, Model *mod, float huberDelta,
                             const dartk_da_point *d_pts, int nElements, const float *d_A, const int *d_map,
                             float *d_result, hipStream_t st) {
    const ModelView &m = mod->v;
    const int d = RED ? redDims : fullDims;
    DARTK_REQUIRE(fullDims == 6 + m.nJoints, "dims %d does not match the model's 6+%d", fullDims, m.nJoints);
    DARTK_REQUIRE(d >= 6 && d <= kMaxDims, "system dimension %d out of range [6,%d]", d, kMaxDims);
    DARTK_REQUIRE(nElements >= 0, "negative element count");
    const int S = sys_size(d);
    if (nElements <= 10) {    // obsToMod.cu:644-646: memset only
        DARTK_HIP(hipMemsetAsync(d_result, 0, S * sizeof(float), st));
        return DARTK_OK;
    }
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const int nTiles = (nElements + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocks = nTiles < kMaxPartialBlocks ? nTiles : kMaxPartialBlocks;
    rc = ensure_partials(ws, (size_t)nBlocks * S);
    if (rc) return rc;
    const size_t ldsBytes = ne_obs2mod_lds_bytes(m, d, RED);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
    const float4 *vert = reinterpret_cast<const float4 *>(d_vert);
#define DARTK_LAUNCH_NE(PPT_)                                                                                   \
    do {                                                                                                        \
        auto kfn = ne_obs2mod_kernel<PPT_, RED>;                                                                \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks), dim3(NE_BLOCK), ldsBytes, st, d_pts, vert, nElements, m,         \
                           huberDelta, redDims, d_A, d_map, lane_partials(ws), (const int *)nullptr, FrameBatch{1, 0, 0, 0, 0, 0, 0, 0}); \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_NE(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_NE(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_NE(6);
    else DARTK_LAUNCH_NE(20);
#undef DARTK_LAUNCH_NE
    DARTK_HIP(hipGetLastError());
    return launch_reduce_slabs(lane_partials(ws), nBlocks, S, d_result, 1, st);
}

// the same normal equations with the list length read on the device (*d_count, at most maxElements): a fixed grid walks
// the tiles, and the "more than 10 elements" gate of the wrapper (obsToMod.cu:644-646) is applied by the reduction
template <int RED>
static int launch_ne_obs2mod_dev(int fullDims, int redDims, const float *d_vert, Model *mod, float huberDelta,
                                 const dartk_da_point *d_pts, const int *d_count, int maxElements, const float *d_A,
                                 const int *d_map, float *d_result, hipStream_t st) {
    const ModelView &m = mod->v;
    const int d = RED ? redDims : fullDims;
    DARTK_REQUIRE(d >= 6 && d <= kMaxDims, "system dimension %d out of range [6,%d]", d, kMaxDims);
    const int S = sys_size(d);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const FrameBatch fb = cur_batch();
    const int nTiles = (maxElements + NE_BLOCK - 1) / NE_BLOCK;
    const int cap = fb.n > 1 ? 128 : 512;           // a batch fills the chip with its frames: fewer, longer workgroups per frame
    const int nBlocks = nTiles < cap ? (nTiles > 0 ? nTiles : 1) : cap;
    rc = ensure_partials(ws, (size_t)fb.n * nBlocks * S);
    if (rc) return rc;
    const size_t ldsBytes = ne_obs2mod_lds_bytes(m, d, RED);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
    const float4 *vert = reinterpret_cast<const float4 *>(d_vert);
#define DARTK_LAUNCH_NE(PPT_)                                                                                   \
    do {                                                                                                        \
        auto kfn = ne_obs2mod_kernel<PPT_, RED>;                                                                \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks, fb.n), dim3(NE_BLOCK), ldsBytes, st, d_pts, vert, maxElements, m, \
                           huberDelta, redDims, d_A, d_map, lane_partials(ws), d_count, fb);                         \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_NE(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_NE(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_NE(6);
    else DARTK_LAUNCH_NE(20);
#undef DARTK_LAUNCH_NE
    DARTK_HIP(hipGetLastError());
    return launch_reduce_slabs_gated(lane_partials(ws), nBlocks, S, d_result, d_count, 10, st);
}

int gn_launch_ne_obs2mod_dev(int red, int fullDims, int redDims, const float *d_vert, Model *mod, float huberDelta,
                             const dartk_da_point *d_pts, const int *d_count, int maxElements, const float *d_A,
                             const int *d_map, float *d_result, hipStream_t st) {
    if (red == 1) return launch_ne_obs2mod_dev<1>(fullDims, redDims, d_vert, mod, huberDelta, d_pts, d_count, maxElements, d_A, nullptr, d_result, st);
    if (red == 2) return launch_ne_obs2mod_dev<2>(fullDims, redDims, d_vert, mod, huberDelta, d_pts, d_count, maxElements, nullptr, d_map, d_result, st);
    return launch_ne_obs2mod_dev<0>(fullDims, fullDims, d_vert, mod, huberDelta, d_pts, d_count, maxElements, nullptr, nullptr, d_result, st);
}

// -----------------------------------------------------------------------------------------------------
// The same normal equations for callers that cannot allocate (code under HIP graph capture, tight serving loops):
// dartk_ne_obs2mod_noalloc. The partial-system slabs -- the one scratch buffer of this term, which dartk_ne_obs2mod takes
// from the library's hipMalloc'ed workspace -- live in a file-scope device array of a fixed, modest size (kNoAllocMaxBlocks
// slabs of a system of up to kNoAllocMaxDims dimensions = 561 KiB per GPU, held until the process ends). `scratch` itself is a
// host-side stand-in for the symbol: the device address is looked up once per device (hipGetSymbolAddress, in dartk_init; the
// entry point refuses a device dartk_init has not seen) and only that pointer is ever handed to kernels. Nothing has to be
// cleared between calls: every launched workgroup overwrites its whole slab and the reduction reads exactly the slabs of
// this launch. Calls that overlap on one GPU share the array and race.
// -----------------------------------------------------------------------------------------------------
static constexpr int kNoAllocMaxDims = 32;
static constexpr int kNoAllocMaxBlocks = 256;
static constexpr size_t kNoAllocFloats = (size_t)kNoAllocMaxBlocks * sys_size(kNoAllocMaxDims);
static __device__ __align__(256) float scratch[kNoAllocFloats];
static float *g_scratchPtr[16] = {nullptr};       // per device: each GPU holds its own copy of the array at its own address

// dartk_init(device): resolves this device's address of `scratch` (the device must be current); no allocation
int noalloc_scratch_resolve() {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    if (!g_scratchPtr[dev]) {
        float *ptr = nullptr;
        DARTK_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        DARTK_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the no-allocation scratch array");
        g_scratchPtr[dev] = ptr;
    }
    return DARTK_OK;
}

// the entry point's side: the address dartk_init resolved for the current device, or an error -- nothing is initialised lazily
// inside the call (dartk_init creates the library's workspace, streams and events, which allocates)
static int noalloc_scratch(float **out) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16 && g_scratchPtr[dev],
                  "dartk_ne_obs2mod_noalloc: dartk_init(%d) has not run on this device (the scratch address is resolved there, not inside the call)", dev);
    *out = g_scratchPtr[dev];
    return DARTK_OK;
}

static int launch_ne_obs2mod_noalloc(int dims, const float *d_vert, Model *mod, float huberDelta, const dartk_da_point *d_pts,
                                     int nElements, float *d_result, hipStream_t st) {
    const ModelView &m = mod->v;
    DARTK_REQUIRE(dims <= kNoAllocMaxDims, "dartk_ne_obs2mod_noalloc supports at most %d dimensions (fixed scratch array), got %d; use dartk_ne_obs2mod",
                  kNoAllocMaxDims, dims);
    DARTK_REQUIRE(dims == 6 + m.nJoints, "dims %d does not match the model's 6+%d", dims, m.nJoints);
    DARTK_REQUIRE(nElements >= 0, "negative element count");
    const int S = sys_size(dims);
    if (nElements <= 10) {    // obsToMod.cu:644-646: memset only
        DARTK_HIP(hipMemsetAsync(d_result, 0, S * sizeof(float), st));
        return DARTK_OK;
    }
    float *slabs = nullptr;
    int rc = noalloc_scratch(&slabs);
    if (rc) return rc;
    const int nTiles = (nElements + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocks = nTiles < kNoAllocMaxBlocks ? nTiles : kNoAllocMaxBlocks;      // the kernel strides over the tiles
    const size_t ldsBytes = ne_obs2mod_lds_bytes(m, dims, 0);
    DARTK_REQUIRE(ldsBytes <= 64 * 1024, "LDS requirement %zu exceeds the 64 KiB a launch gets without an attribute call", ldsBytes);
    const float4 *vert = reinterpret_cast<const float4 *>(d_vert);
    const FrameBatch one{1, 0, 0, 0, 0, 0, 0, 0};
    // the instantiation dartk_ne_obs2mod picks for this system size (launch_ne_obs2mod)
#define DARTK_LAUNCH_NA(PPT_)                                                                                             \
    hipLaunchKernelGGL((ne_obs2mod_kernel<PPT_, 0>), dim3(nBlocks), dim3(NE_BLOCK), ldsBytes, st, d_pts, vert, nElements, m, huberDelta, \
                       dims, (const float *)nullptr, (const int *)nullptr, slabs, (const int *)nullptr, one)
    if (S <= NE_BLOCK) DARTK_LAUNCH_NA(1);
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_NA(3);
    else DARTK_LAUNCH_NA(6);
#undef DARTK_LAUNCH_NA
    DARTK_HIP(hipGetLastError());
    // straight to the reduction kernel: launch_reduce_slabs would consult the workspace (which get_workspace creates on first
    // use) and a deferred-reduction table that belongs to another call
    return launch_reduce_slabs_direct(slabs, nBlocks, S, d_result, st);
}

// multi-model association for the device-resident loop: the argument block is uploaded once per optimizePoses call
// (the ModelViews point at pose arrays that are updated in place), then one launch per iteration
int gn_assoc_multi_prepare(int nModels, Model *const *mods, const float *distThr, const float *planeOffset, const float *pn3,
                           dartk_da_point *const *pts, void **d_args_out, hipStream_t st, size_t *stageBytes) {
    if (stageBytes) {
        size_t w = 0;
        for (int i = 0; i < nModels; ++i) w += (size_t)assoc_stage_words(mods[i]->v);
        *stageBytes = w * sizeof(float);
    }
    ArgSlot slot;           // pinned staging: the upload needs no host wait (the slot is released as soon as the copy is enqueued)
    int rc = arg_ring_acquire(sizeof(MultiArgs), &slot);
    if (rc) return rc;
    MultiArgs &hargs = *reinterpret_cast<MultiArgs *>(slot.h);
    hargs.nModels = nModels;
    for (int i = 0; i < nModels; ++i) {
        hargs.m[i] = mods[i]->v;
        hargs.distThr[i] = distThr[i];
        hargs.planeOffset[i] = planeOffset[i];
        for (int k = 0; k < 3; ++k) hargs.pn[i][k] = pn3[3 * i + k];
        hargs.pts[i] = pts[i];
    }
    MultiArgs *d_args = nullptr;
    rc = get_multi_args_buffer((void **)&d_args, sizeof(MultiArgs));
    if (rc) return rc;
    DARTK_HIP(hipMemcpyAsync(d_args, &hargs, sizeof(MultiArgs), hipMemcpyHostToDevice, st));
    rc = arg_ring_release(slot, st);
    if (rc) return rc;
    *d_args_out = d_args;
    return DARTK_OK;
}
int gn_assoc_multi_launch(const float *d_vert, int nPix, int nModels, const void *d_args, int *d_last, hipStream_t st,
                          const int *d_validIdx, const int *d_nValid, size_t stageBytes, bool zeroCounts) {
    const FrameBatch fb = cur_batch();
    if (zeroCounts) DARTK_HIP(hipMemsetAsync(d_last, 0, (fb.n > 1 ? (size_t)fb.n * fb.counts : (size_t)nModels) * sizeof(int), st));
    int wgs = (nPix + 255) / 256;
    if (d_validIdx) { const int cap = fb.n > 1 ? (8192 / fb.n > 16 ? 8192 / fb.n : 16) : 2048; if (wgs > cap) wgs = cap; }
    // with a valid-pixel list the grid is chip-sized and each workgroup walks many pixels: the tables are worth staging
    if (d_validIdx && stageBytes > 0 && stageBytes <= 64 * 1024 && !getenv("DARTK_ASSOC_UNSTAGED"))
        hipLaunchKernelGGL((assoc_multi_kernel<false, true>), dim3(wgs, fb.n), dim3(256), stageBytes, st, (const float4 *)d_vert, nPix,
                           (const MultiArgs *)d_args, d_last, nullptr, nullptr, fb, d_validIdx, d_nValid);
    else
        hipLaunchKernelGGL((assoc_multi_kernel<false, false>), dim3(wgs, fb.n), dim3(256), 0, st, (const float4 *)d_vert, nPix,
                           (const MultiArgs *)d_args, d_last, nullptr, nullptr, fb, d_validIdx, d_nValid);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// dartk_assoc_multi; d_T_mcs (device, [nModels][12], may be null) replaces the models' own camera -> model transforms for this
// launch only, as the reference's kernel argument does (obsToMod.h:88-107): the models are not touched
int gn_assoc_multi_entry(const float *d_vert, const float *d_norm, int width, int height, int nModels, const dartk_model_t *models,
                         const float *d_T_mcs, const float *h_distThresholds, const float *h_planeOffsets, const float *h_planeNormals3,
                         dartk_da_point *const *h_pts_table, int32_t *d_last, int32_t *d_dbgDA, float *d_dbgErr, void *stream) {
    (void)d_norm;
    DARTK_REQUIRE(d_vert && models && h_distThresholds && h_planeOffsets && h_planeNormals3 && h_pts_table && d_last,
                  "dartk_assoc_multi: null argument");
    DARTK_REQUIRE(nModels >= 1 && nModels <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", nModels, DARTK_MAX_MODELS);
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    hipStream_t st = as_stream(stream);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    // the argument block goes through the ring of pinned slots: no host wait per call (the reference's wrapper does not block either)
    ArgSlot slot;
    rc = arg_ring_acquire(sizeof(MultiArgs), &slot);
    if (rc) return rc;
    MultiArgs &hargs = *reinterpret_cast<MultiArgs *>(slot.h);
    hargs.nModels = nModels;
    for (int i = 0; i < nModels; ++i) {
        Model *mod = reinterpret_cast<Model *>(models[i]);
        DARTK_REQUIRE(mod && h_pts_table[i], "dartk_assoc_multi: null model or point buffer %d", i);
        hargs.m[i] = mod->v;
        if (d_T_mcs) hargs.m[i].T_mc = const_cast<float *>(d_T_mcs) + 12 * i;
        hargs.distThr[i] = h_distThresholds[i];
        hargs.planeOffset[i] = h_planeOffsets[i];
        for (int k = 0; k < 3; ++k) hargs.pn[i][k] = h_planeNormals3[3 * i + k];
        hargs.pts[i] = h_pts_table[i];
    }
    MultiArgs *d_args = reinterpret_cast<MultiArgs *>(slot.d);
    DARTK_HIP(hipMemcpyAsync(d_args, &hargs, sizeof(MultiArgs), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipMemsetAsync(d_last, 0, nModels * sizeof(int), st));
    const int nPix = width * height;
    const dim3 grid((nPix + 255) / 256), block(256);
    if (d_dbgDA || d_dbgErr)
        hipLaunchKernelGGL(assoc_multi_kernel<true>, grid, block, 0, st, (const float4 *)d_vert, nPix, d_args, d_last, d_dbgDA, d_dbgErr, FrameBatch{1, 0, 0, 0, 0, 0, 0, 0},
                           (const int *)nullptr, (const int *)nullptr);
    else
        hipLaunchKernelGGL(assoc_multi_kernel<false>, grid, block, 0, st, (const float4 *)d_vert, nPix, d_args, d_last, nullptr, nullptr, FrameBatch{1, 0, 0, 0, 0, 0, 0, 0},
                           (const int *)nullptr, (const int *)nullptr);
    DARTK_HIP(hipGetLastError());
    return arg_ring_release(slot, st);
}

DARTK_COUNTER_ACCESS(obs2mod)

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_assoc_single(const float *d_vert, const float *d_norm, int width, int height, dartk_model_t h,
                       float distThreshold, float normThreshold, float planeOffset, const float *pn,
                       dartk_da_point *d_pts, int32_t *d_last, int32_t *h_last, int32_t *d_dbgDA, float *d_dbgErr,
                       float *d_dbgNorm, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_norm && d_pts && d_last && pn, "dartk_assoc_single: null argument");
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    hipStream_t st = as_stream(stream);
    const int nPix = width * height;
    DARTK_HIP(hipMemsetAsync(d_last, 0, sizeof(int), st));
    const dim3 grid((nPix + 255) / 256), block(256);
    const bool dbg = d_dbgDA || d_dbgErr || d_dbgNorm;
    if (dbg)
        hipLaunchKernelGGL(assoc_single_kernel<true>, grid, block, 0, st, (const float4 *)d_vert, (const float4 *)d_norm,
                           nPix, mod->v, distThreshold, normThreshold, planeOffset, pn[0], pn[1], pn[2], d_last, d_pts,
                           d_dbgDA, d_dbgErr, (float4 *)d_dbgNorm);
    else
        hipLaunchKernelGGL(assoc_single_kernel<false>, grid, block, 0, st, (const float4 *)d_vert, (const float4 *)d_norm,
                           nPix, mod->v, distThreshold, normThreshold, planeOffset, pn[0], pn[1], pn[2], d_last, d_pts,
                           nullptr, nullptr, nullptr);
    DARTK_HIP(hipGetLastError());
    if (h_last) {
        DARTK_HIP(hipMemcpyAsync(h_last, d_last, sizeof(int), hipMemcpyDeviceToHost, st));
        DARTK_HIP(hipStreamSynchronize(st));
    }
    return DARTK_OK;
}

int dartk_assoc_multi(const float *d_vert, const float *d_norm, int width, int height, int nModels,
                      const dartk_model_t *models, const float *h_distThresholds, const float *h_planeOffsets,
                      const float *h_planeNormals3, dartk_da_point *const *h_pts_table, int32_t *d_last,
                      int32_t *d_dbgDA, float *d_dbgErr, void *stream) {
    return gn_assoc_multi_entry(d_vert, d_norm, width, height, nModels, models, nullptr, h_distThresholds, h_planeOffsets, h_planeNormals3,
                                h_pts_table, d_last, d_dbgDA, d_dbgErr, stream);
}

int dartk_ne_obs2mod(int dims, const float *d_vert, int width, int height, dartk_model_t h, float huberDelta,
                     const dartk_da_point *d_pts, int nElements, float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_pts && d_result, "dartk_ne_obs2mod: null argument");
    DARTK_REQUIRE(nElements <= width * height, "nElements %d exceeds the map size", nElements);
    return launch_ne_obs2mod<0>(dims, dims, d_vert, mod, huberDelta, d_pts, nElements, nullptr, nullptr, d_result, as_stream(stream));
}

int dartk_ne_obs2mod_noalloc(int dims, const float *d_vert, int width, int height, dartk_model_t h, float huberDelta,
                             const dartk_da_point *d_pts, int nElements, float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_pts && d_result, "dartk_ne_obs2mod_noalloc: null argument");
    DARTK_REQUIRE(width > 0 && height > 0 && nElements <= width * height, "nElements %d exceeds the map size", nElements);
    return launch_ne_obs2mod_noalloc(dims, d_vert, mod, huberDelta, d_pts, nElements, d_result, as_stream(stream));
}

int dartk_ne_obs2mod_reduced(int dims, int redDims, const float *d_A, const float *d_vert, int width, int height,
                             dartk_model_t h, float huberDelta, const dartk_da_point *d_pts, int nElements,
                             float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_pts && d_result && d_A, "dartk_ne_obs2mod_reduced: null argument");
    DARTK_REQUIRE(nElements <= width * height, "nElements %d exceeds the map size", nElements);
    DARTK_REQUIRE(redDims >= 6 && redDims <= dims, "redDims %d must be in [6,%d]", redDims, dims);
    return launch_ne_obs2mod<1>(dims, redDims, d_vert, mod, huberDelta, d_pts, nElements, d_A, nullptr, d_result, as_stream(stream));
}

int dartk_ne_obs2mod_parammap(int dims, int redDims, const int32_t *d_map, const float *d_vert, int width, int height,
                              dartk_model_t h, float huberDelta, const dartk_da_point *d_pts, int nElements,
                              float *d_result, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_pts && d_result && d_map, "dartk_ne_obs2mod_parammap: null argument");
    DARTK_REQUIRE(nElements <= width * height, "nElements %d exceeds the map size", nElements);
    DARTK_REQUIRE(redDims >= 6 && redDims <= dims, "redDims %d must be in [6,%d]", redDims, dims);
    return launch_ne_obs2mod<2>(dims, redDims, d_vert, mod, huberDelta, d_pts, nElements, nullptr, d_map, d_result, as_stream(stream));
}

}  // extern "C"
