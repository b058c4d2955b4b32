// dartk_internal.h -- device-side data layout and helpers shared by the dartk HIP translation units.
//
// HBM layout of one model (a single arena, see dartk_model.hip):
//   [static]  SdfDev sdfs[nSdfs] | int sdfFrame[] | u64 depMask[nFrames] | int jointType[] | float jointAxis[3*nJ]
//             | float T_pf[12*nJ] | float dh[4*nJ] | float limits[2*nJ] | int frameParent[] | int dependencies[]
//   [pose]    float T_mc[12] | T_cm[12] | T_mf[12*nFrames] | T_fm[12*nFrames] | joints[nJ]
//   [grids]   all SDF voxel data back to back (voxel units, x fastest, 256-B aligned per grid) | obs SDF
// Kernels receive a ModelView by value (pointers into the arena). Indices that are wave-uniform (the loop
// over SDFs, the loop over joints) are read through the scalar cache; per-lane-divergent transform lookups
// (the associated SDF's frame) are served from an LDS copy staged once per workgroup.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dartk.h"

namespace dartk {

struct SdfDev {               // 64 bytes
    int dimx, dimy, dimz;
    int dataOff;              // offset (in floats) of this grid in the model's grid arena
    float ox, oy, oz, res;
    // x / res without the IEEE division sequence (~12 instructions): q0 = x*rcpRes, q = fma(fma(-q0, res, x), rcpRes, q0).
    // divExact is set at model creation only if that equals the correctly rounded x / res for EVERY fp32 significand x
    // (exhaustive check of the 2^23 significands for this res; the result is exponent independent), else the kernels divide.
    float rcpRes;
    int divExact;
    // exact float images of the reference's double-literal bound tests (grid_3d.h:63-80):
    //   p > 2.50001  <=>  p >= lo25 ;   p < dim - 2.50001  <=>  p <= hi25{x,y,z}
    float lo25, hi25x, hi25y, hi25z;
    float lo05, hi05x, hi05y, hi05z;
    // "keep boxes" in grid coordinates: a trilinear lookup at g reads voxels x0 = (int)(g.x-0.5) and x0+1 (same for y, z)
    // and is a convex combination of them, so it can only be < 0 (box 0) or <= kNearThr[k-1] metres (boxes 1..3) if that
    // 2x2x2 cell touches the bounding box of the voxels that are. Outside the box the lookup cannot change any decision
    // of the association (threshold test), the intersection terms (r < 0) or the contact priors (|err| <= threshold), so
    // it is skipped: identical results, far fewer gathers than the reference's padded grids imply (padding 10 cm).
    float keepLo[4][3], keepHi[4][3];       // keep iff keepLo <= g < keepHi on every axis
    int nbx, nby;             // bricks per row / per slab (brick layout only, see DARTK_SDF_LAYOUT)
};
static constexpr int kSdfWords = (int)(sizeof(SdfDev) / 4);
static_assert(sizeof(SdfDev) % 4 == 0, "SdfDev is copied word by word");
static constexpr float kNearThr[3] = {0.02f, 0.03f, 0.05f};
// box to use for a distance threshold (metres): the tightest one that still covers it; -1 = no pruning
__host__ __device__ inline int keep_slot(float thr) {
    return (thr <= kNearThr[0]) ? 1 : (thr <= kNearThr[1]) ? 2 : (thr <= kNearThr[2]) ? 3 : -1;
}

struct ModelView {
    int nFrames, nJoints, nSdfs, modelId;
    const SdfDev *sdfs;
    const int *sdfFrame;
    const unsigned long long *depMask;
    const int *jointType;
    const float *jointAxis;
    const float *T_pf;
    const float *dh;
    const float *limits;
    const int *frameParent;
    const float *grids;
    float *T_mc;      // [12] camera -> model
    float *T_cm;      // [12]
    float *T_mf;      // [12*nFrames] frame -> model
    float *T_fm;      // [12*nFrames] model -> frame
    float *joints;    // [nJoints] (clamped)
    SdfDev obs;       // observation SDF header (dataOff unused)
    float *obsData;
    // Batches of independent frames (dartk_optimize_poses_batch): the pose arrays above point at frame 0 of per-frame records
    // [T_mc | T_cm | T_mf | T_fm | joints] that lie poseStride floats apart, the observation SDF at frame 0 of grids obsStride
    // floats apart; kernels launched with grid.y = frames call select_frame(). Both are 0 for a model handle's own pose.
    int poseStride, obsStride;
};

struct Model {        // host-side handle (dartk_model_t)
    ModelView v;
    void *arena;
    size_t arenaBytes;
    int device;
    // host copies needed by wrappers
    float *h_pin;     // pinned staging for pose uploads
    size_t obsVoxels;
    float *dtTmp;     // obs-SDF distance transform scratch (generateObsSdf)
    bool uniformRes;  // all SDFs share one resolution
};

// ---- storage layout of the model SDFs in HBM (the reference: dense, x fastest, grid_3d.h:82-84) ---------------------------
//   0  dense: data[x + dimx*(y + dimy*z)]; a trilinear lookup is four 8-byte loads on four rows dimx*4 / dimx*dimy*4 bytes apart
//   1  4x4x4 bricks with a one-voxel apron: brick (bx,by,bz) stores voxels [4b, 4b+4] per axis as 5x5x5 floats in a 512-byte
//      slot; the eight corners of any lookup lie in ONE brick (four 8-byte loads within 512 contiguous bytes); 2x the memory
//   2  one 32-byte record per cell: the eight corners of cell (x0,y0,z0) side by side, one aligned 32-byte fetch; 8x the memory
// Values and interpolation arithmetic are identical in all three (bit-exact results); chosen at build time, measured in
// DESIGN.md "SDF layout".
#ifndef DARTK_SDF_LAYOUT
#define DARTK_SDF_LAYOUT 1      // measured equal to 0 and 2 within noise (profiles/r02_sdf_layouts.txt); 1 is the layout north_star names
#endif
inline size_t sdf_layout_floats(int dx, int dy, int dz) {
#if DARTK_SDF_LAYOUT == 1
    return (size_t)((dx + 3) / 4) * ((dy + 3) / 4) * ((dz + 3) / 4) * 128;
#elif DARTK_SDF_LAYOUT == 2
    return (size_t)dx * dy * dz * 8;
#else
    return (size_t)dx * dy * dz;
#endif
}

// ---- error handling -------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define DARTK_HIP(call)                                                                      \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) {                                                              \
            dartk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DARTK_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)
#define DARTK_REQUIRE(cond, ...)                                                             \
    do {                                                                                     \
        if (!(cond)) { dartk::set_error(__VA_ARGS__); return DARTK_ERR_INVALID; }            \
    } while (0)

// one deferred slab reduction (see Workspace::deferReduce)
struct ReduceJob {
    const float *slabs;
    float *result;
    const int *count;       // gate: keep the sum only if *count > minCount (null: `keep`)
    int nBlocks, S, keep, minCount;
};
static constexpr int kMaxReduceJobs = 48;

// per-device workspace: partial-system slabs for the deterministic two-pass reductions
struct Workspace {
    float *partials;       // [maxBlocks][maxSys]
    size_t partialFloats;
    int *counters;         // small int scratch
    float *sysScratch;     // packed systems (device GN loop)
    float *rayBuf;         // ray-cast traces: [nSdfs][nPix] {detection t, hit t}
    size_t rayFloats;
    int *validIdx;         // compacted indices of valid observation pixels + block offsets
    size_t validInts;
    hipStream_t auxStream; // second stream: the mod->obs chain runs beside the obs->mod chain (fork/join by events)
    int partialLane;       // which third of `partials` the next term wrapper uses (device loops run three chains side by side)
    hipStream_t auxStream2; // third stream: the per-call obs-SDF build runs beside the first ray-cast
    hipEvent_t evFork, evJoin, evObs, evJoin2, evTrace;
    // The multi-model loop's term wrappers each end in a slab reduction of their own (seven small launches per iteration, and
    // that loop is bound by the host's launches). deferReduce: the wrappers only note their reduction (reduceJobs) and leave
    // their slabs at deferOffset[lane] of their lane; ONE launch after the chains have joined reduces all of them
    // (flush_reduce_jobs). measureLanes: a dry pass that adds up what a lane would need (laneNeed), so that the slab buffer can
    // be grown BEFORE slabs that must survive until the flush are in it.
    bool deferReduce, measureLanes;
    int nReduceJobs;
    size_t deferOffset[3], laneNeed[3];
    ReduceJob reduceJobs[kMaxReduceJobs];
};
static constexpr int kPartialLanes = 3;
// the slab buffer of the current lane; partialFloats is the capacity of ONE lane
inline float *lane_partials(Workspace *w) {
    return w->partials + (size_t)w->partialLane * w->partialFloats + (w->deferReduce ? w->deferOffset[w->partialLane] : 0);
}
int flush_reduce_jobs(Workspace *w, hipStream_t st);
int get_workspace(Workspace **ws);
static constexpr int kMaxPartialBlocks = 2048;
static constexpr int kMaxDims = 70;                       // 6 + DARTK_MAX_JOINTS
static constexpr int kMaxSys = (kMaxDims + 1) * (kMaxDims + 2) / 2;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// every translation unit with SDF lookups defines lookup_counters_<name>(out[8], reset) through this macro (see CNT_*)
#ifdef DARTK_COUNT_LOOKUPS
#define DARTK_COUNTER_ACCESS(name)                                                                           \
    int lookup_counters_##name(unsigned long long *out, int reset) {                                          \
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                    \
        DARTK_HIP(hipDeviceSynchronize());                                                                    \
        DARTK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lookupCount), sizeof(h)));                              \
        for (int i = 0; i < 8; ++i) { out[i] += h[i]; h[i] = 0; }                                             \
        if (reset) DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lookupCount), h, sizeof(h)));                     \
        return DARTK_OK;                                                                                      \
    }
#else
#define DARTK_COUNTER_ACCESS(name) int lookup_counters_##name(unsigned long long *, int) { return DARTK_OK; }
#endif
int lookup_counters_obs2mod(unsigned long long *out, int reset);
int lookup_counters_gn(unsigned long long *out, int reset);
int lookup_counters_mod2obs(unsigned long long *out, int reset);
int lookup_counters_intersection(unsigned long long *out, int reset);
int lookup_counters_noobs(unsigned long long *out, int reset);

// ---- device helpers -------------------------------------------------------------------------------
#ifdef __HIPCC__

// rebase a (local copy of a) ModelView on frame f of a batch (no-op for f == 0 or a plain model handle)
__device__ __forceinline__ void select_frame(ModelView &m, int f) {
    const size_t o = (size_t)f * (size_t)m.poseStride;
    m.T_mc += o; m.T_cm += o; m.T_mf += o; m.T_fm += o; m.joints += o;
    m.obsData += (size_t)f * (size_t)m.obsStride;
}

struct f3 { float x, y, z; };

// SE3Transform(A, float4 b) with b.w carried (SE3.h:411-416): keeps the reference's left-to-right sums.
__device__ __forceinline__ f3 se3_apply_w(const float *A, float bx, float by, float bz, float bw) {
    f3 o;
    o.x = A[0] * bx + A[1] * by + A[2] * bz + A[3] * bw;
    o.y = A[4] * bx + A[5] * by + A[6] * bz + A[7] * bw;
    o.z = A[8] * bx + A[9] * by + A[10] * bz + A[11] * bw;
    return o;
}
// SE3Transform(A, float3 b) (SE3.h:435-439)
__device__ __forceinline__ f3 se3_apply(const float *A, f3 b) {
    f3 o;
    o.x = A[0] * b.x + A[1] * b.y + A[2] * b.z + A[3];
    o.y = A[4] * b.x + A[5] * b.y + A[6] * b.z + A[7];
    o.z = A[8] * b.x + A[9] * b.y + A[10] * b.z + A[11];
    return o;
}
// SE3Rotate (SE3.h:441-445)
__device__ __forceinline__ f3 se3_rotate(const float *A, f3 b) {
    f3 o;
    o.x = A[0] * b.x + A[1] * b.y + A[2] * b.z;
    o.y = A[4] * b.x + A[5] * b.y + A[6] * b.z;
    o.z = A[8] * b.x + A[9] * b.y + A[10] * b.z;
    return o;
}
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    f3 o;
    o.x = a.y * b.z - a.z * b.y;
    o.y = a.z * b.x - a.x * b.z;
    o.z = a.x * b.y - a.y * b.x;
    return o;
}
// SE3Transform(A,B) (SE3.h:396-409) and SE3Invert (:447-451) on float[12]
__device__ __forceinline__ void se3_mul(const float *A, const float *B, float *o) {
    float r[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a0 = A[4 * i], a1 = A[4 * i + 1], a2 = A[4 * i + 2], a3 = A[4 * i + 3];
        r[4 * i + 0] = a0 * B[0] + a1 * B[4] + a2 * B[8];
        r[4 * i + 1] = a0 * B[1] + a1 * B[5] + a2 * B[9];
        r[4 * i + 2] = a0 * B[2] + a1 * B[6] + a2 * B[10];
        r[4 * i + 3] = a0 * B[3] + a1 * B[7] + a2 * B[11] + a3;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = r[i];
}
__device__ __forceinline__ void se3_invert(const float *A, float *o) {
    float r[12];
    r[0] = A[0]; r[1] = A[4]; r[2] = A[8];   r[3]  = -(A[0] * A[3] + A[4] * A[7] + A[8] * A[11]);
    r[4] = A[1]; r[5] = A[5]; r[6] = A[9];   r[7]  = -(A[1] * A[3] + A[5] * A[7] + A[9] * A[11]);
    r[8] = A[2]; r[9] = A[6]; r[10] = A[10]; r[11] = -(A[2] * A[3] + A[6] * A[7] + A[10] * A[11]);
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = r[i];
}

// ---- Grid3D lookups (grid_3d.h). All of this is exact fp32: for 0.5 <= p < 2^23 the reference's double
// expressions (p - 0.5), (int)(p - 0.5) and the fraction are exactly representable in fp32, so evaluating
// them in fp32 gives bit-identical x0 / fx (DESIGN.md "mixed precision"); the double-literal bound tests
// are replaced by their exact float images precomputed on the host.
// x / res, correctly rounded, for a res whose reciprocal sequence was verified (SdfDev::divExact)
__device__ __forceinline__ float div_exact(float x, float res, float rcp) {
    const float q0 = x * rcp;
    return __builtin_fmaf(__builtin_fmaf(-q0, res, x), rcp, q0);
}
__device__ __forceinline__ f3 grid_coords(const SdfDev &s, f3 p) {        // grid_3d.h:49-51
    const float res = s.res, rcp = s.rcpRes;
    const float dx = p.x - s.ox, dy = p.y - s.oy, dz = p.z - s.oz;
    f3 g;
    if (s.divExact) { g.x = div_exact(dx, res, rcp); g.y = div_exact(dy, res, rcp); g.z = div_exact(dz, res, rcp); }
    else { g.x = dx / res; g.y = dy / res; g.z = dz / res; }
    return g;
}
__device__ __forceinline__ bool in_bounds_grad_interp(const SdfDev &s, f3 g) {  // grid_3d.h:76-80
    return g.x >= s.lo25 && g.x <= s.hi25x && g.y >= s.lo25 && g.y <= s.hi25y && g.z >= s.lo25 && g.z <= s.hi25z;
}
__device__ __forceinline__ bool in_bounds_interp(const SdfDev &s, f3 g) {       // grid_3d.h:63-67
    return g.x >= s.lo05 && g.x <= s.hi05x && g.y >= s.lo05 && g.y <= s.hi05y && g.z >= s.lo05 && g.z <= s.hi05z;
}
__device__ __forceinline__ bool in_keep_box(const SdfDev &s, int slot, f3 g) {
    if (slot < 0) return true;
    return g.x >= s.keepLo[slot][0] && g.x < s.keepHi[slot][0] && g.y >= s.keepLo[slot][1] && g.y < s.keepHi[slot][1] &&
           g.z >= s.keepLo[slot][2] && g.z < s.keepHi[slot][2];
}
__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + t * (b - a); }
struct __attribute__((packed, aligned(4))) f2u { float a, b; };   // x-neighbours fetched with one 8-byte load

// ---- lookup counters (diagnostic build -DDARTK_COUNT_LOOKUPS = libdartk_count.so only; bench.py's roofline leg prices the
// trilinear lookups the kernels PERFORM, not the ones the reference's un-pruned loops imply). One array per translation unit
// (no relocatable device code), summed by dartk_lookup_counters(). Categories follow the oracle's (oracle/dart_oracle.h).
enum { CNT_OTHER = 0, CNT_ASSOC = 1, CNT_ASSOC_GRAD = 2, CNT_NE_O2M = 3, CNT_RAYCAST = 4, CNT_M2O = 5, CNT_INTER = 6, CNT_PRIOR = 7, CNT_N = 8 };
#ifdef DARTK_COUNT_LOOKUPS
static __device__ unsigned long long g_lookupCount[CNT_N];
template <int CAT> __device__ __forceinline__ void count_lookup() { if constexpr (CAT >= 0) atomicAdd(&g_lookupCount[CAT], 1ull); }   // CAT < 0: not counted
#else
template <int CAT> __device__ __forceinline__ void count_lookup() {}
#endif

// getValueInterpolated (grid_3d.h:86-112); data points at voxel (0,0,0) of the grid
template <int CAT = CNT_OTHER>
__device__ __forceinline__ float grid_value(const float *__restrict__ data, int dimx, int dimy, f3 g) {
    count_lookup<CAT>();
    const float tx = g.x - 0.5f, ty = g.y - 0.5f, tz = g.z - 0.5f;
    const int x0 = (int)tx, y0 = (int)ty, z0 = (int)tz;
    const float fx = tx - (float)x0, fy = ty - (float)y0, fz = tz - (float)z0;
    const float *p = data + (x0 + dimx * (y0 + dimy * z0));
    const int sy = dimx, sz = dimx * dimy;
    const f2u q00 = *reinterpret_cast<const f2u *>(p), q10 = *reinterpret_cast<const f2u *>(p + sy);
    const f2u q01 = *reinterpret_cast<const f2u *>(p + sz), q11 = *reinterpret_cast<const f2u *>(p + sz + sy);
    const float v000 = q00.a, v100 = q00.b;
    const float v010 = q10.a, v110 = q10.b;
    const float v001 = q01.a, v101 = q01.b;
    const float v011 = q11.a, v111 = q11.b;
    const float dx00 = lerpf(v000, v100, fx);
    const float dx01 = lerpf(v001, v101, fx);
    const float dx10 = lerpf(v010, v110, fx);
    const float dx11 = lerpf(v011, v111, fx);
    const float dxy0 = lerpf(dx00, dx10, fy);
    const float dxy1 = lerpf(dx01, dx11, fy);
    return lerpf(dxy0, dxy1, fz);
}
// getGradientInterpolated (grid_3d.h:314-330): six interpolated lookups at +-1 voxel, 0.5*(f+ - f-)
template <int CAT = CNT_OTHER>
__device__ __forceinline__ f3 grid_gradient(const float *__restrict__ data, int dimx, int dimy, f3 g) {
    f3 q, r;
    q = g; q.x = g.x + 1.0f; const float fpx = grid_value<CAT>(data, dimx, dimy, q);
    q = g; q.y = g.y + 1.0f; const float fpy = grid_value<CAT>(data, dimx, dimy, q);
    q = g; q.z = g.z + 1.0f; const float fpz = grid_value<CAT>(data, dimx, dimy, q);
    q = g; q.x = g.x - 1.0f; const float fmx = grid_value<CAT>(data, dimx, dimy, q);
    q = g; q.y = g.y - 1.0f; const float fmy = grid_value<CAT>(data, dimx, dimy, q);
    q = g; q.z = g.z - 1.0f; const float fmz = grid_value<CAT>(data, dimx, dimy, q);
    r.x = 0.5f * (fpx - fmx);
    r.y = 0.5f * (fpy - fmy);
    r.z = 0.5f * (fpz - fmz);
    return r;
}

// lookups in one of a model's SDFs (grids = the model's grid arena), in the build's storage layout (DARTK_SDF_LAYOUT below):
// the VALUES and the interpolation arithmetic are those of grid_value(); only where the eight corners are fetched from differs
template <int CAT = CNT_OTHER>
__device__ __forceinline__ float sdf_lookup(const float *__restrict__ data, int dimx, int dimy, int nbx, int nby, f3 g) {
#if DARTK_SDF_LAYOUT == 0
    (void)nbx; (void)nby;
    return grid_value<CAT>(data, dimx, dimy, g);
#else
    count_lookup<CAT>();
    const float tx = g.x - 0.5f, ty = g.y - 0.5f, tz = g.z - 0.5f;
    const int x0 = (int)tx, y0 = (int)ty, z0 = (int)tz;
    const float fx = tx - (float)x0, fy = ty - (float)y0, fz = tz - (float)z0;
#if DARTK_SDF_LAYOUT == 1
    (void)dimx; (void)dimy;
    const float *p = data + (size_t)(((z0 >> 2) * nby + (y0 >> 2)) * nbx + (x0 >> 2)) * 128 + ((x0 & 3) + 5 * ((y0 & 3) + 5 * (z0 & 3)));
    const f2u q00 = *reinterpret_cast<const f2u *>(p), q10 = *reinterpret_cast<const f2u *>(p + 5);
    const f2u q01 = *reinterpret_cast<const f2u *>(p + 25), q11 = *reinterpret_cast<const f2u *>(p + 30);
    const float v000 = q00.a, v100 = q00.b, v010 = q10.a, v110 = q10.b, v001 = q01.a, v101 = q01.b, v011 = q11.a, v111 = q11.b;
#else
    (void)nbx; (void)nby;
    const float4 *p = reinterpret_cast<const float4 *>(data) + 2 * (size_t)(x0 + dimx * (y0 + dimy * z0));
    const float4 lo = p[0], hi = p[1];
    const float v000 = lo.x, v100 = lo.y, v010 = lo.z, v110 = lo.w, v001 = hi.x, v101 = hi.y, v011 = hi.z, v111 = hi.w;
#endif
    const float dx00 = lerpf(v000, v100, fx);
    const float dx01 = lerpf(v001, v101, fx);
    const float dx10 = lerpf(v010, v110, fx);
    const float dx11 = lerpf(v011, v111, fx);
    const float dxy0 = lerpf(dx00, dx10, fy);
    const float dxy1 = lerpf(dx01, dx11, fy);
    return lerpf(dxy0, dxy1, fz);
#endif
}
template <int CAT = CNT_OTHER>
__device__ __forceinline__ float sdf_value(const float *__restrict__ grids, const SdfDev &s, f3 g) {
    return sdf_lookup<CAT>(grids + s.dataOff, s.dimx, s.dimy, s.nbx, s.nby, g);
}
template <int CAT = CNT_OTHER>
__device__ __forceinline__ f3 sdf_gradient(const float *__restrict__ grids, const SdfDev &s, f3 g) {
    const float *data = grids + s.dataOff;
    f3 q, r;
    q = g; q.x = g.x + 1.0f; const float fpx = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    q = g; q.y = g.y + 1.0f; const float fpy = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    q = g; q.z = g.z + 1.0f; const float fpz = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    q = g; q.x = g.x - 1.0f; const float fmx = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    q = g; q.y = g.y - 1.0f; const float fmy = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    q = g; q.z = g.z - 1.0f; const float fmz = sdf_lookup<CAT>(data, s.dimx, s.dimy, s.nbx, s.nby, q);
    r.x = 0.5f * (fpx - fmx);
    r.y = 0.5f * (fpy - fmy);
    r.z = 0.5f * (fpz - fmz);
    return r;
}
// projectToSdfSurface (sdf.cpp:6-23) in grid coordinates: steps of the interpolated distance along the normalised interpolated
// gradient until |distance| < threshold (voxel units), the point leaves the gradient-interpolation bounds, or maxIters steps.
// Not inlined into its callers: it runs for the rare contact priors that move their contact point.
template <int CAT = CNT_OTHER>
__device__ __noinline__ f3 project_to_sdf_surface(const float *__restrict__ grids, const SdfDev &s, f3 g, float threshold, int maxIters) {
    for (int iter = 0; iter < maxIters; ++iter) {
        if (!in_bounds_grad_interp(s, g)) return g;
        const float dist = sdf_value<CAT>(grids, s, g);
        if (fabsf(dist) < threshold) return g;
        const f3 grad = sdf_gradient<CAT>(grids, s, g);
        const float invLen = 1.0f / sqrtf(grad.x * grad.x + grad.y * grad.y + grad.z * grad.z);      // helper_math normalize on the host
        g.x = g.x - dist * (grad.x * invLen);
        g.y = g.y - dist * (grad.y * invLen);
        g.z = g.z - dist * (grad.z * invLen);
    }
    return g;
}


// ---- articulated Jacobian (kernel_common.h:12-71) ---------------------------------------------------
// Writes J[0..dims) (or the nJoints articulation entries when ARTIC_ONLY) with stride 1 into `J`.
// T_fm / T_mf may point to LDS or global memory. depMask = this frame's kinematic-chain bit mask.
template <bool ARTIC_ONLY>
__device__ __forceinline__ void error_jacobian(float *J, f3 pm, float pw, unsigned long long depMask, f3 grad,
                                               int nJoints, const int *__restrict__ jointType,
                                               const float *__restrict__ jointAxis, const float *T_fm,
                                               const float *T_mf) {
    float *Ja = J;
    if (!ARTIC_ONLY) {
        J[0] = grad.x * -1.0f + grad.y * 0.0f + grad.z * 0.0f;
        J[1] = grad.x * 0.0f + grad.y * -1.0f + grad.z * 0.0f;
        J[2] = grad.x * 0.0f + grad.y * 0.0f + grad.z * -1.0f;
        J[3] = grad.x * 0.0f + grad.y * pm.z + grad.z * -pm.y;
        J[4] = grad.x * -pm.z + grad.y * 0.0f + grad.z * pm.x;
        J[5] = grad.x * pm.y + grad.y * -pm.x + grad.z * 0.0f;
        Ja = J + 6;
    }
    for (int i = 0; i < nJoints; ++i) {
        float v = 0.0f;
        if ((depMask >> i) & 1ull) {
            const int jf = i + 1;
            const float *Tm = T_mf + 12 * jf;
            f3 ax; ax.x = jointAxis[3 * i]; ax.y = jointAxis[3 * i + 1]; ax.z = jointAxis[3 * i + 2];
            if (jointType[i] == 0) {
                const f3 rel = se3_apply_w(T_fm + 12 * jf, pm.x, pm.y, pm.z, pw);
                const f3 dxa = cross(ax, rel);
                const f3 dx = se3_rotate(Tm, dxa);
                v = dot(grad, dx);
            } else {
                const f3 axm = se3_rotate(Tm, ax);
                v = dot(grad, axm);
            }
        }
        Ja[i] = v;
    }
}

// The same entries with each lane walking only the joints of ITS chain (a hand's finger link depends on 4 of 16 joints; the
// lanes of a wave sit on different fingers, so the loop above runs the full body for nearly every joint). jointType /
// jointAxis are indexed per lane here: pass LDS copies. Entry for entry the arithmetic of error_jacobian.
template <bool ARTIC_ONLY, bool ZERO = true>
__device__ __forceinline__ void error_jacobian_walk(float *J, f3 pm, float pw, unsigned long long depMask, f3 grad, int nJoints,
                                                    const int *sJointType, const float *sJointAxis, const float *T_fm,
                                                    const float *T_mf) {
    float *Ja = J;
    if (!ARTIC_ONLY) {
        J[0] = grad.x * -1.0f + grad.y * 0.0f + grad.z * 0.0f;
        J[1] = grad.x * 0.0f + grad.y * -1.0f + grad.z * 0.0f;
        J[2] = grad.x * 0.0f + grad.y * 0.0f + grad.z * -1.0f;
        J[3] = grad.x * 0.0f + grad.y * pm.z + grad.z * -pm.y;
        J[4] = grad.x * -pm.z + grad.y * 0.0f + grad.z * pm.x;
        J[5] = grad.x * pm.y + grad.y * -pm.x + grad.z * 0.0f;
        Ja = J + 6;
    }
    if (ZERO)       // (false: the caller has cleared the row)
        for (int i = 0; i < nJoints; ++i) Ja[i] = 0.0f;
    unsigned long long todo = nJoints < 64 ? depMask & ((1ull << nJoints) - 1ull) : depMask;
    while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int jf = i + 1;
        const float *Tm = T_mf + 12 * jf;
        f3 ax; ax.x = sJointAxis[3 * i]; ax.y = sJointAxis[3 * i + 1]; ax.z = sJointAxis[3 * i + 2];
        float v;
        if (sJointType[i] == 0) {
            const f3 rel = se3_apply_w(T_fm + 12 * jf, pm.x, pm.y, pm.z, pw);
            const f3 dxa = cross(ax, rel);
            const f3 dx = se3_rotate(Tm, dxa);
            v = dot(grad, dx);
        } else {
            const f3 axm = se3_rotate(Tm, ax);
            v = dot(grad, axm);
        }
        Ja[i] = v;
    }
}

// The same entries again with the row shared by the 4 neighbouring lanes q = 0..3 of one wave: entry k of J (pose columns and
// joints alike) is written by the lane with ((k + phase) & 3) == q and by no other, so a row costs a quarter of the walk's
// latency when few rows are pending. Entry for entry the arithmetic of error_jacobian; ZERO clears this lane's joints first.
template <bool ARTIC_ONLY, bool ZERO>
__device__ __forceinline__ void error_jacobian_walk4(float *J, f3 pm, float pw, unsigned long long depMask, f3 grad, int nJoints,
                                                     const int *sJointType, const float *sJointAxis, const float *T_fm,
                                                     const float *T_mf, int q, int phase) {
    int aOff = 0;
    if (!ARTIC_ONLY) {
        for (int k = (q - phase) & 3; k < 6; k += 4) {
            float v;
            switch (k) {
                case 0: v = grad.x * -1.0f + grad.y * 0.0f + grad.z * 0.0f; break;
                case 1: v = grad.x * 0.0f + grad.y * -1.0f + grad.z * 0.0f; break;
                case 2: v = grad.x * 0.0f + grad.y * 0.0f + grad.z * -1.0f; break;
                case 3: v = grad.x * 0.0f + grad.y * pm.z + grad.z * -pm.y; break;
                case 4: v = grad.x * -pm.z + grad.y * 0.0f + grad.z * pm.x; break;
                default: v = grad.x * pm.y + grad.y * -pm.x + grad.z * 0.0f; break;
            }
            J[k] = v;
        }
        aOff = 6;
    }
    float *Ja = J + aOff;
    const int first = (q - phase - aOff) & 3;      // this lane's joints: first, first + 4, ..
    if (ZERO)
        for (int i = first; i < nJoints; i += 4) Ja[i] = 0.0f;
    unsigned long long todo = nJoints < 64 ? depMask & ((1ull << nJoints) - 1ull) : depMask;
    todo &= 0x1111111111111111ull << first;
    while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int jf = i + 1;
        const float *Tm = T_mf + 12 * jf;
        f3 ax; ax.x = sJointAxis[3 * i]; ax.y = sJointAxis[3 * i + 1]; ax.z = sJointAxis[3 * i + 2];
        float v;
        if (sJointType[i] == 0) {
            const f3 rel = se3_apply_w(T_fm + 12 * jf, pm.x, pm.y, pm.z, pw);
            const f3 dxa = cross(ax, rel);
            const f3 dx = se3_rotate(Tm, dxa);
            v = dot(grad, dx);
        } else {
            const f3 axm = se3_rotate(Tm, ax);
            v = dot(grad, axm);
        }
        Ja[i] = v;
    }
}

#endif  // __HIPCC__

}  // namespace dartk
