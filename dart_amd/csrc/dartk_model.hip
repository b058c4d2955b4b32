// dartk_model.hip -- library state, device buffers, model upload and on-device forward kinematics.
// Replaces: util/mirrored_memory.h (device half), model/mirrored_model.cpp:8-225.
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_fk.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <vector>
#include <utility>

namespace dartk {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local FrameBatch t_batch = {1, 0, 0, 0, 0, 0, 0, 0};
FrameBatch &cur_batch() { return t_batch; }

static Workspace g_ws[16];
static bool g_ws_ready[16] = {false};

int get_workspace(Workspace **out) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    if (!g_ws_ready[dev]) {
        Workspace &w = g_ws[dev];
        w.partialFloats = (size_t)kMaxPartialBlocks * 1024;   // grown on demand by ensure_partials
        w.partialLane = 0;
        DARTK_HIP(hipMalloc((void **)&w.partials, kPartialLanes * w.partialFloats * sizeof(float)));
        DARTK_HIP(hipMalloc((void **)&w.counters, 64 * sizeof(int)));
        DARTK_HIP(hipMalloc((void **)&w.sysScratch, (size_t)8 * kMaxSys * sizeof(float)));
        w.rayBuf = nullptr; w.rayFloats = 0;
        w.validIdx = nullptr; w.validInts = 0;
        DARTK_HIP(hipStreamCreateWithFlags(&w.auxStream, hipStreamNonBlocking));
        DARTK_HIP(hipEventCreateWithFlags(&w.evFork, hipEventDisableTiming));
        DARTK_HIP(hipEventCreateWithFlags(&w.evJoin, hipEventDisableTiming));
        DARTK_HIP(hipStreamCreateWithFlags(&w.auxStream2, hipStreamNonBlocking));
        DARTK_HIP(hipEventCreateWithFlags(&w.evObs, hipEventDisableTiming));
        DARTK_HIP(hipEventCreateWithFlags(&w.evJoin2, hipEventDisableTiming));
        DARTK_HIP(hipEventCreateWithFlags(&w.evTrace, hipEventDisableTiming));
        g_ws_ready[dev] = true;
    }
    *out = &g_ws[dev];
    return DARTK_OK;
}

// Argument blocks that are too large to pass by value travel through a small ring of (pinned host, device) slot pairs: a call
// fills the next slot, uploads it asynchronously and marks it released once its last consumer is enqueued. The host only ever
// waits when the ring has wrapped around onto a slot whose consumer has not finished -- eight calls ahead of the device.
static constexpr int kArgSlots = 8;
static constexpr size_t kArgSlotBytes = 65536;
size_t arg_slot_bytes() { return kArgSlotBytes; }
struct ArgRing { char *d; char *h; hipEvent_t ev[kArgSlots]; bool used[kArgSlots]; int next; bool ready; };
static ArgRing g_argRing[16];
int arg_ring_acquire(size_t bytes, ArgSlot *out) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    DARTK_REQUIRE(bytes <= kArgSlotBytes, "argument block of %zu bytes exceeds the %zu-byte slots", bytes, kArgSlotBytes);
    ArgRing &r = g_argRing[dev];
    if (!r.ready) {
        DARTK_HIP(hipMalloc((void **)&r.d, kArgSlots * kArgSlotBytes));
        DARTK_HIP(hipHostMalloc((void **)&r.h, kArgSlots * kArgSlotBytes, hipHostMallocDefault));
        for (int i = 0; i < kArgSlots; ++i) { DARTK_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming)); r.used[i] = false; }
        r.next = 0; r.ready = true;
    }
    const int i = r.next;
    r.next = (r.next + 1) % kArgSlots;
    if (r.used[i]) DARTK_HIP(hipEventSynchronize(r.ev[i]));
    r.used[i] = false;
    out->d = r.d + (size_t)i * kArgSlotBytes; out->h = r.h + (size_t)i * kArgSlotBytes; out->idx = i; out->dev = dev;
    return DARTK_OK;
}
// slot -> device memory by a kernel that reads the pinned slot (mapped into the device's address space). hipMemcpyAsync holds the
// HOST until the stream has drained for copies beyond a few KB (measured on ROCm 7.2: the 34 KB of launch tables of the
// multi-model loop kept every call in step with the device, 680 us per call); a kernel launch never waits.
__global__ void __launch_bounds__(256)
upload_words_kernel(unsigned *__restrict__ dst, const unsigned *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int arg_ring_upload(void *d_dst, const ArgSlot &s, size_t bytes, hipStream_t st) {
    DARTK_REQUIRE(bytes % 4 == 0 && bytes <= kArgSlotBytes, "arg_ring_upload: %zu bytes", bytes);
    const size_t n = bytes / 4;
    const unsigned grid = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    hipLaunchKernelGGL(upload_words_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, (unsigned *)d_dst, (const unsigned *)s.h, n);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
int arg_ring_release(const ArgSlot &s, hipStream_t st) {
    ArgRing &r = g_argRing[s.dev];
    DARTK_HIP(hipEventRecord(r.ev[s.idx], st));
    r.used[s.idx] = true;
    return DARTK_OK;
}

static void *g_multiArgs[16] = {nullptr};
static size_t g_multiArgsBytes[16] = {0};
int get_multi_args_buffer(void **d_ptr, size_t bytes) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    if (g_multiArgsBytes[dev] < bytes) {
        if (g_multiArgs[dev]) DARTK_HIP(hipFree(g_multiArgs[dev]));
        DARTK_HIP(hipMalloc(&g_multiArgs[dev], bytes));
        g_multiArgsBytes[dev] = bytes;
    }
    *d_ptr = g_multiArgs[dev];
    return DARTK_OK;
}

int ensure_partials(Workspace *w, size_t floats) {
    const size_t aligned = (floats + 63) / 64 * 64;
    if (w->measureLanes) w->laneNeed[w->partialLane] += aligned;
    if (w->deferReduce) {       // the lane already holds slabs that wait for the flush: it cannot be re-allocated now
        DARTK_REQUIRE(w->deferOffset[w->partialLane] + floats <= w->partialFloats, "deferred reductions: slab lane %d overflows (%zu + %zu > %zu floats)",
                      w->partialLane, w->deferOffset[w->partialLane], floats, w->partialFloats);
        return DARTK_OK;
    }
    if (floats <= w->partialFloats) return DARTK_OK;
    DARTK_HIP(hipDeviceSynchronize());
    DARTK_HIP(hipFree(w->partials));
    w->partialFloats = floats;          // per lane
    DARTK_HIP(hipMalloc((void **)&w->partials, kPartialLanes * floats * sizeof(float)));
    return DARTK_OK;
}

int ensure_ray_buf(Workspace *w, size_t floats) {
    if (floats <= w->rayFloats) return DARTK_OK;
    DARTK_HIP(hipDeviceSynchronize());
    if (w->rayBuf) DARTK_HIP(hipFree(w->rayBuf));
    w->rayBuf = nullptr; w->rayFloats = 0;
    DARTK_HIP(hipMalloc((void **)&w->rayBuf, floats * sizeof(float)));
    w->rayFloats = floats;
    return DARTK_OK;
}

int ensure_valid_idx(Workspace *w, size_t ints) {
    if (ints <= w->validInts) return DARTK_OK;
    DARTK_HIP(hipDeviceSynchronize());
    if (w->validIdx) DARTK_HIP(hipFree(w->validIdx));
    w->validIdx = nullptr; w->validInts = 0;
    DARTK_HIP(hipMalloc((void **)&w->validIdx, ints * sizeof(int)));
    w->validInts = ints;
    return DARTK_OK;
}

// 1024 threads = 32 outputs x 32 segments: segment g sums slabs g, g+32, ... in fp64 (fixed order), the 32 segment
// sums are then added in segment order by the first 32 threads; a wave reads two 128-byte rows per step
static constexpr int kReduceOuts = 32;
static constexpr int kReduceSegs = 32;

__global__ void __launch_bounds__(kReduceOuts * kReduceSegs)
reduce_slabs_kernel(const float *__restrict__ slabs, int nBlocks, int S, float *__restrict__ result, int keep,
                    const int *__restrict__ count, int minCount, FrameBatch fb) {
    __shared__ double part[kReduceSegs][kReduceOuts + 1];
    slabs += (size_t)blockIdx.y * nBlocks * S;      // frame blockIdx.y of a batch (0 otherwise)
    result += (size_t)blockIdx.y * fb.sys;
    if (count) count += (size_t)blockIdx.y * fb.counts;
    const int col = threadIdx.x % kReduceOuts, seg = threadIdx.x / kReduceOuts;
    const int o = blockIdx.x * kReduceOuts + col;
    if (count) keep = *count > minCount;
    double s = 0.0;
    if (keep && o < S) {
        int b = seg;
        // (sixteen loads in flight, added in slab order: the sums are those of the plain loop, the chain of round trips a quarter
        // of what four at a time leave -- this kernel's duration is that chain)
        for (; b + 15 * kReduceSegs < nBlocks; b += 16 * kReduceSegs) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = slabs[(size_t)(b + k * kReduceSegs) * S + o];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += (double)v[k];
        }
        for (; b + 3 * kReduceSegs < nBlocks; b += 4 * kReduceSegs) {
            const float v0 = slabs[(size_t)b * S + o];
            const float v1 = slabs[(size_t)(b + kReduceSegs) * S + o];
            const float v2 = slabs[(size_t)(b + 2 * kReduceSegs) * S + o];
            const float v3 = slabs[(size_t)(b + 3 * kReduceSegs) * S + o];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; b < nBlocks; b += kReduceSegs) s += (double)slabs[(size_t)b * S + o];
    }
    part[seg][col] = s;
    __syncthreads();
    if (seg == 0 && o < S) {
        double t = 0.0;
        for (int g = 0; g < kReduceSegs; ++g) t += part[g][col];
        result[o] = (float)t;
    }
}

// every noted reduction in one launch: grid (outputs / 32, frame, job)
struct ReduceJobTable { int n; ReduceJob job[kMaxReduceJobs]; };
__global__ void __launch_bounds__(kReduceOuts * kReduceSegs)
reduce_slabs_jobs_kernel(ReduceJobTable t, FrameBatch fb) {
    __shared__ double part[kReduceSegs][kReduceOuts + 1];
    const ReduceJob &j = t.job[blockIdx.z];
    const int S = j.S, nBlocks = j.nBlocks;
    if ((int)(blockIdx.x * kReduceOuts) >= S) return;
    const float *__restrict__ slabs = j.slabs + (size_t)blockIdx.y * nBlocks * S;
    float *__restrict__ result = j.result + (size_t)blockIdx.y * fb.sys;
    const int col = threadIdx.x % kReduceOuts, seg = threadIdx.x / kReduceOuts;
    const int o = blockIdx.x * kReduceOuts + col;
    const int keep = j.count ? (j.count[(size_t)blockIdx.y * fb.counts] > j.minCount) : j.keep;
    double s = 0.0;
    if (keep && o < S) {        // the sums of reduce_slabs_kernel, term for term
        int b = seg;
        // (sixteen loads in flight, added in slab order: the sums are those of the plain loop, the chain of round trips a quarter
        // of what four at a time leave -- this kernel's duration is that chain)
        for (; b + 15 * kReduceSegs < nBlocks; b += 16 * kReduceSegs) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = slabs[(size_t)(b + k * kReduceSegs) * S + o];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += (double)v[k];
        }
        for (; b + 3 * kReduceSegs < nBlocks; b += 4 * kReduceSegs) {
            const float v0 = slabs[(size_t)b * S + o];
            const float v1 = slabs[(size_t)(b + kReduceSegs) * S + o];
            const float v2 = slabs[(size_t)(b + 2 * kReduceSegs) * S + o];
            const float v3 = slabs[(size_t)(b + 3 * kReduceSegs) * S + o];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; b < nBlocks; b += kReduceSegs) s += (double)slabs[(size_t)b * S + o];
    }
    part[seg][col] = s;
    __syncthreads();
    if (seg == 0 && o < S) {
        double tt = 0.0;
        for (int g = 0; g < kReduceSegs; ++g) tt += part[g][col];
        result[o] = (float)tt;
    }
}

static bool defer_reduce(const float *slabs, int nBlocks, int S, float *result, int keep, const int *count, int minCount, int *rc) {
    Workspace *w;
    if (get_workspace(&w) != DARTK_OK || !w->deferReduce) return false;
    *rc = DARTK_OK;
    if (w->nReduceJobs >= kMaxReduceJobs) { set_error("more than %d deferred reductions in one iteration", kMaxReduceJobs); *rc = DARTK_ERR_INVALID; return true; }
    w->reduceJobs[w->nReduceJobs++] = ReduceJob{slabs, result, count, nBlocks, S, keep, minCount};
    w->deferOffset[w->partialLane] += ((size_t)cur_batch().n * nBlocks * S + 63) / 64 * 64;
    return true;
}

int flush_reduce_jobs(Workspace *w, hipStream_t st) {
    if (w->nReduceJobs > 0) {
        ReduceJobTable t;
        t.n = w->nReduceJobs;
        int maxS = 1;
        for (int i = 0; i < t.n; ++i) { t.job[i] = w->reduceJobs[i]; if (t.job[i].S > maxS) maxS = t.job[i].S; }
        hipLaunchKernelGGL(reduce_slabs_jobs_kernel, dim3((maxS + kReduceOuts - 1) / kReduceOuts, cur_batch().n, t.n), dim3(kReduceOuts * kReduceSegs),
                           0, st, t, cur_batch());
        DARTK_HIP(hipGetLastError());
    }
    w->nReduceJobs = 0;
    for (int l = 0; l < kPartialLanes; ++l) w->deferOffset[l] = 0;
    return DARTK_OK;
}

int launch_reduce_slabs(const float *slabs, int nBlocks, int S, float *result, int keep, hipStream_t stream) {
    int drc;
    if (defer_reduce(slabs, nBlocks, S, result, keep, nullptr, 0, &drc)) return drc;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((S + kReduceOuts - 1) / kReduceOuts, cur_batch().n), dim3(kReduceOuts * kReduceSegs), 0,
                       stream, slabs, nBlocks, S, result, keep, (const int *)nullptr, 0, cur_batch());
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// one frame, now, on `stream`: no workspace, no deferred-job table, no batch scope (dartk_ne_obs2mod_noalloc)
int launch_reduce_slabs_direct(const float *slabs, int nBlocks, int S, float *result, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((S + kReduceOuts - 1) / kReduceOuts, 1), dim3(kReduceOuts * kReduceSegs), 0, stream,
                       slabs, nBlocks, S, result, 1, (const int *)nullptr, 0, FrameBatch{1, 0, 0, 0, 0, 0, 0, 0});
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int launch_reduce_slabs_gated(const float *slabs, int nBlocks, int S, float *result, const int *d_count, int minCount,
                              hipStream_t stream) {
    int drc;
    if (defer_reduce(slabs, nBlocks, S, result, 0, d_count, minCount, &drc)) return drc;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((S + kReduceOuts - 1) / kReduceOuts, cur_batch().n), dim3(kReduceOuts * kReduceSegs), 0,
                       stream, slabs, nBlocks, S, result, 0, d_count, minCount, cur_batch());
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// The pose travels in the kernel ARGUMENTS (copied by the launch itself): no staging buffer whose reuse would need the host
// to wait for the stream, no host->device copy packets; dartk_model_set_pose is one asynchronous launch.
struct PoseArgs { int hasT, nJ; float T_mc[12]; float joints[DARTK_MAX_JOINTS]; };
__global__ void __launch_bounds__(64) fk_kernel(ModelView m, int mode, PoseArgs p) {
    __shared__ float sT[12 * (DARTK_MAX_JOINTS + 1)];
    if (p.hasT && threadIdx.x < 12) m.T_mc[threadIdx.x] = p.T_mc[threadIdx.x];
    if ((int)threadIdx.x < p.nJ) m.joints[threadIdx.x] = p.joints[threadIdx.x];      // lane j writes joint j and is the lane fk_wave reads it with
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mode == 0 && threadIdx.x == 0) {
        float T[12], inv[12];
        for (int i = 0; i < 12; ++i) T[i] = p.hasT ? p.T_mc[i] : m.T_mc[i];
        se3_invert(T, inv);
        for (int i = 0; i < 12; ++i) m.T_cm[i] = inv[i];
    }
    fk_wave(m, mode, sT);
}

}  // namespace dartk

using namespace dartk;

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char *dartk_last_error(void) { return g_err; }
int dartk_version(void) { return 1; }

int dartk_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) { set_error("no HIP device available (%s)", hipGetErrorString(e)); return DARTK_ERR_NO_DEVICE; }
    DARTK_REQUIRE(device >= 0 && device < n, "device %d out of range (%d devices)", device, n);
    DARTK_HIP(hipSetDevice(device));
    Workspace *w;
    int rc = get_workspace(&w);
    if (rc) return rc;
    return noalloc_scratch_resolve();       // resolve this device's address of the no-allocation scratch array once
}

int dartk_buf_alloc(void **d_ptr, size_t bytes) {
    DARTK_REQUIRE(d_ptr != nullptr, "dartk_buf_alloc: null out pointer");
    DARTK_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
    return DARTK_OK;
}
int dartk_buf_free(void *d_ptr) { DARTK_HIP(hipFree(d_ptr)); return DARTK_OK; }
int dartk_buf_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    DARTK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    DARTK_HIP(hipStreamSynchronize(as_stream(stream)));
    return DARTK_OK;
}
int dartk_buf_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    DARTK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    DARTK_HIP(hipStreamSynchronize(as_stream(stream)));
    return DARTK_OK;
}
int dartk_buf_zero(void *d_ptr, size_t bytes, void *stream) {
    DARTK_HIP(hipMemsetAsync(d_ptr, 0, bytes, as_stream(stream)));
    return DARTK_OK;
}
int dartk_stream_sync(void *stream) { DARTK_HIP(hipStreamSynchronize(as_stream(stream))); return DARTK_OK; }

static float next_above(double v) {   // smallest float strictly greater than v
    float f = (float)v;
    while ((double)f <= v) f = nextafterf(f, INFINITY);
    while ((double)nextafterf(f, -INFINITY) > v) f = nextafterf(f, -INFINITY);
    return f;
}
static float next_below(double v) {   // largest float strictly less than v
    float f = (float)v;
    while ((double)f >= v) f = nextafterf(f, -INFINITY);
    while ((double)nextafterf(f, INFINITY) < v) f = nextafterf(f, INFINITY);
    return f;
}
// true iff q = fma(fma(-(x*r), c, x), r, x*r) with r = RN(1/c) equals RN(x / c) for every fp32 significand x in [1, 2)
// (both sides scale exactly with the exponent of x as long as nothing is subnormal). ~20 ms per distinct c, memoised.
static bool division_by_constant_is_exact(float c) {
    static std::vector<std::pair<float, bool>> memo;
    for (const auto &e : memo) if (e.first == c) return e.second;
    bool ok = c > 0 && std::isfinite(c) && std::isnormal(1.0f / c);
    if (ok) {
        const float r = 1.0f / c;
        for (uint32_t m = 0; m < (1u << 23) && ok; ++m) {
            uint32_t bits = 0x3f800000u | m;
            float x;
            memcpy(&x, &bits, 4);
            const float q0 = x * r;
            const float q = std::fmaf(std::fmaf(-q0, c, x), r, q0);
            ok = (q == x / c);
        }
    }
    memo.emplace_back(c, ok);
    return ok;
}

static void fill_bounds(SdfDev &s) {
    s.rcpRes = s.res > 0 ? 1.0f / s.res : 0.0f;
    // DARTK_NO_FASTDIV=1 (diagnostic): always divide
    static const bool noFast = getenv("DARTK_NO_FASTDIV") != nullptr;
    s.divExact = (!noFast && s.res > 0 && division_by_constant_is_exact(s.res)) ? 1 : 0;
    s.lo25 = next_above(2.50001);
    s.hi25x = next_below((double)(unsigned)s.dimx - 2.50001);
    s.hi25y = next_below((double)(unsigned)s.dimy - 2.50001);
    s.hi25z = next_below((double)(unsigned)s.dimz - 2.50001);
    s.lo05 = next_above(0.50001);
    s.hi05x = next_below((double)(unsigned)s.dimx - 0.50001);
    s.hi05y = next_below((double)(unsigned)s.dimy - 0.50001);
    s.hi05z = next_below((double)(unsigned)s.dimz - 0.50001);
}

// bounding boxes (grid coordinates) of the voxels with value < 0 and with value*res <= kNearThr[k]; see SdfDev
static void fill_keep_boxes(SdfDev &s, const float *data) {
    int lo[4][3], hi[4][3];
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 3; ++a) { lo[k][a] = 1 << 30; hi[k][a] = -1; }
    for (int z = 0; z < s.dimz; ++z) for (int y = 0; y < s.dimy; ++y) {
        const float *row = data + (size_t)s.dimx * (y + (size_t)s.dimy * z);
        for (int x = 0; x < s.dimx; ++x) {
            const float v = row[x];
            // compared in metres as the kernels do (v*res)
            const float d = v * s.res;
            // small safety margins: an interpolated value may round a few ulp outside the range of its corners
            const bool in[4] = {!(v >= 1e-4f), !(d > kNearThr[0] * 1.0001f), !(d > kNearThr[1] * 1.0001f), !(d > kNearThr[2] * 1.0001f)};
            const int c[3] = {x, y, z};
            for (int k = 0; k < 4; ++k) if (in[k]) for (int a = 0; a < 3; ++a) { if (c[a] < lo[k][a]) lo[k][a] = c[a]; if (c[a] > hi[k][a]) hi[k][a] = c[a]; }
        }
    }
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 3; ++a) {
        if (hi[k][a] < 0) { s.keepLo[k][a] = 1e30f; s.keepHi[k][a] = -1e30f; }
        else { s.keepLo[k][a] = (float)lo[k][a] - 0.5f; s.keepHi[k][a] = (float)hi[k][a] + 1.5f; }
    }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int dartk_model_create(const dartk_model_desc *d, dartk_model_t *out) {
    DARTK_REQUIRE(d && out, "dartk_model_create: null argument");
    DARTK_REQUIRE(d->n_frames >= 1 && d->n_joints == d->n_frames - 1,
                  "model must have n_joints == n_frames-1 (joint i drives frame i+1, mirrored_model.h:38); got %d/%d",
                  d->n_joints, d->n_frames);
    DARTK_REQUIRE(d->n_joints <= DARTK_MAX_JOINTS, "at most %d joints supported, got %d", DARTK_MAX_JOINTS, d->n_joints);
    DARTK_REQUIRE(d->n_sdfs >= 1 && d->n_sdfs <= 65535, "n_sdfs %d out of range", d->n_sdfs);
    for (int s = 0; s < d->n_sdfs; ++s) {
        DARTK_REQUIRE(d->sdf_frame[s] >= 0 && d->sdf_frame[s] < d->n_frames, "sdf %d: frame %d out of range", s, d->sdf_frame[s]);
        DARTK_REQUIRE(d->sdf_dim[3 * s] >= 6 && d->sdf_dim[3 * s + 1] >= 6 && d->sdf_dim[3 * s + 2] >= 6,
                      "sdf %d: every dimension must be >= 6 voxels", s);
        DARTK_REQUIRE(d->sdf_resolution[s] > 0, "sdf %d: resolution must be positive", s);
    }
    for (int f = 1; f < d->n_frames; ++f)
        DARTK_REQUIRE(d->frame_parent[f] >= 0 && d->frame_parent[f] < f, "frame %d: parent %d must precede it", f, d->frame_parent[f]);

    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;

    const int nF = d->n_frames, nJ = d->n_joints, nS = d->n_sdfs;
    const int nJa = nJ > 0 ? nJ : 1;
    // arena layout
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_sdfs = take(sizeof(SdfDev) * nS);
    const size_t o_sdfFrame = take(sizeof(int) * nS);
    const size_t o_dep = take(sizeof(unsigned long long) * nF);
    const size_t o_jt = take(sizeof(int) * nJa);
    const size_t o_ax = take(sizeof(float) * 3 * nJa);
    const size_t o_tpf = take(sizeof(float) * 12 * nJa);
    const size_t o_dh = take(sizeof(float) * 4 * nJa);
    const size_t o_lim = take(sizeof(float) * 2 * nJa);
    const size_t o_par = take(sizeof(int) * nF);
    const size_t staticBytes = off;
    const size_t o_Tmc = take(sizeof(float) * 12);
    const size_t o_Tcm = take(sizeof(float) * 12);
    const size_t o_Tmf = take(sizeof(float) * 12 * nF);
    const size_t o_Tfm = take(sizeof(float) * 12 * nF);
    const size_t o_joints = take(sizeof(float) * nJa);
    const size_t o_grids = off;

    std::vector<SdfDev> sdfs(nS);
    size_t gridFloats = 0;
    for (int s = 0; s < nS; ++s) {
        SdfDev &g = sdfs[s];
        g.dimx = d->sdf_dim[3 * s]; g.dimy = d->sdf_dim[3 * s + 1]; g.dimz = d->sdf_dim[3 * s + 2];
        g.ox = d->sdf_offset[3 * s]; g.oy = d->sdf_offset[3 * s + 1]; g.oz = d->sdf_offset[3 * s + 2];
        g.res = d->sdf_resolution[s];
        g.dataOff = (int)gridFloats;
        g.nbx = (g.dimx + 3) / 4; g.nby = (g.dimy + 3) / 4;
        fill_bounds(g);
        fill_keep_boxes(g, d->sdf_data[s]);
        gridFloats = align_up(gridFloats + sdf_layout_floats(g.dimx, g.dimy, g.dimz), 64);
        DARTK_REQUIRE(gridFloats < (size_t)1 << 31, "SDF arena exceeds 2^31 voxels");
    }
    off = align_up(o_grids + gridFloats * sizeof(float), 256);
    const size_t o_obs = off;
    size_t obsVoxels = (size_t)d->obs_sdf_dim[0] * d->obs_sdf_dim[1] * d->obs_sdf_dim[2];
    off = align_up(off + (obsVoxels ? obsVoxels : 1) * sizeof(float), 256);
    const size_t total = off;

    Model *m = new Model();
    memset(m, 0, sizeof(Model));
    struct CreateGuard {        // a failure below frees what was built so far
        Model *m; bool done;
        ~CreateGuard() {
            if (done) return;
            if (m->dtTmp) (void)hipFree(m->dtTmp);
            if (m->h_pin) (void)hipHostFree(m->h_pin);
            if (m->arena) (void)hipFree(m->arena);
            delete m;
        }
    } guard{m, false};
    DARTK_HIP(hipGetDevice(&m->device));
    DARTK_HIP(hipMalloc(&m->arena, total));
    m->arenaBytes = total;
    DARTK_HIP(hipHostMalloc((void **)&m->h_pin, sizeof(float) * (12 + 24 * nF + nJa), hipHostMallocDefault));

    // build the static + grid parts on the host, then one H2D copy
    std::vector<unsigned char> h(total, 0);
    memcpy(h.data() + o_sdfs, sdfs.data(), sizeof(SdfDev) * nS);
    memcpy(h.data() + o_sdfFrame, d->sdf_frame, sizeof(int) * nS);
    unsigned long long *dep = (unsigned long long *)(h.data() + o_dep);
    for (int f = 0; f < nF; ++f) {
        unsigned long long bits = 0;
        for (int j = 0; j < nJ; ++j) if (d->dependencies[f * nJ + j]) bits |= (1ull << j);
        dep[f] = bits;
    }
    if (nJ > 0) {
        memcpy(h.data() + o_jt, d->joint_type, sizeof(int) * nJ);
        memcpy(h.data() + o_ax, d->joint_axis, sizeof(float) * 3 * nJ);
        memcpy(h.data() + o_tpf, d->T_pf, sizeof(float) * 12 * nJ);
        memcpy(h.data() + o_dh, d->dh, sizeof(float) * 4 * nJ);
        memcpy(h.data() + o_lim, d->joint_limits, sizeof(float) * 2 * nJ);
    }
    memcpy(h.data() + o_par, d->frame_parent, sizeof(int) * nF);
    {
        const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(h.data() + o_Tmc, I, sizeof(I));
        memcpy(h.data() + o_Tcm, I, sizeof(I));
        for (int f = 0; f < nF; ++f) {
            memcpy(h.data() + o_Tmf + sizeof(I) * f, I, sizeof(I));
            memcpy(h.data() + o_Tfm + sizeof(I) * f, I, sizeof(I));
        }
    }
    for (int s = 0; s < nS; ++s) {
        const int dx = sdfs[s].dimx, dy = sdfs[s].dimy, dz = sdfs[s].dimz;
        const float *src = d->sdf_data[s];
        float *dst = reinterpret_cast<float *>(h.data() + o_grids) + sdfs[s].dataOff;
        auto at = [&](int x, int y, int z) {      // voxels beyond the grid are never part of a lookup the bound tests admit
            x = x < dx ? x : dx - 1; y = y < dy ? y : dy - 1; z = z < dz ? z : dz - 1;
            return src[x + (size_t)dx * (y + (size_t)dy * z)];
        };
#if DARTK_SDF_LAYOUT == 1
        const int nbx = sdfs[s].nbx, nby = sdfs[s].nby, nbz = (dz + 3) / 4;
        for (int bz = 0; bz < nbz; ++bz) for (int by = 0; by < nby; ++by) for (int bx = 0; bx < nbx; ++bx) {
            float *b = dst + (size_t)((bz * nby + by) * nbx + bx) * 128;
            for (int lz = 0; lz < 5; ++lz) for (int ly = 0; ly < 5; ++ly) for (int lx = 0; lx < 5; ++lx)
                b[lx + 5 * (ly + 5 * lz)] = at(4 * bx + lx, 4 * by + ly, 4 * bz + lz);
        }
#elif DARTK_SDF_LAYOUT == 2
        for (int z = 0; z < dz; ++z) for (int y = 0; y < dy; ++y) for (int x = 0; x < dx; ++x) {
            float *c = dst + 8 * (x + (size_t)dx * (y + (size_t)dy * z));
            for (int k = 0; k < 8; ++k) c[k] = at(x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2));
        }
#else
        (void)at;
        memcpy(dst, src, (size_t)dx * dy * dz * sizeof(float));
#endif
    }
    DARTK_HIP(hipMemcpy(m->arena, h.data(), total, hipMemcpyHostToDevice));
    (void)staticBytes;

    unsigned char *base = (unsigned char *)m->arena;
    ModelView &v = m->v;
    v.nFrames = nF; v.nJoints = nJ; v.nSdfs = nS; v.modelId = d->model_id;
    v.sdfs = (const SdfDev *)(base + o_sdfs);
    v.sdfFrame = (const int *)(base + o_sdfFrame);
    v.depMask = (const unsigned long long *)(base + o_dep);
    v.jointType = (const int *)(base + o_jt);
    v.jointAxis = (const float *)(base + o_ax);
    v.T_pf = (const float *)(base + o_tpf);
    v.dh = (const float *)(base + o_dh);
    v.limits = (const float *)(base + o_lim);
    v.frameParent = (const int *)(base + o_par);
    v.grids = (const float *)(base + o_grids);
    v.T_mc = (float *)(base + o_Tmc);
    v.T_cm = (float *)(base + o_Tcm);
    v.T_mf = (float *)(base + o_Tmf);
    v.T_fm = (float *)(base + o_Tfm);
    v.joints = (float *)(base + o_joints);
    memset(&v.obs, 0, sizeof(SdfDev));
    v.obs.dimx = d->obs_sdf_dim[0]; v.obs.dimy = d->obs_sdf_dim[1]; v.obs.dimz = d->obs_sdf_dim[2];
    v.obs.ox = d->obs_sdf_offset[0]; v.obs.oy = d->obs_sdf_offset[1]; v.obs.oz = d->obs_sdf_offset[2];
    v.obs.res = d->obs_sdf_resolution;
    if (obsVoxels) fill_bounds(v.obs);
    v.obsData = (float *)(base + o_obs);
    m->obsVoxels = obsVoxels;
    m->uniformRes = true;
    for (int s = 1; s < nS; ++s) if (d->sdf_resolution[s] != d->sdf_resolution[0]) m->uniformRes = false;
    m->dtTmp = nullptr;
    if (obsVoxels) DARTK_HIP(hipMalloc((void **)&m->dtTmp, obsVoxels * sizeof(float)));
    guard.done = true;
    registry_add(m);
    *out = reinterpret_cast<dartk_model_t>(m);
    return DARTK_OK;
}

int dartk_model_destroy(dartk_model_t h) {
    Model *m = reinterpret_cast<Model *>(h);
    if (!m) return DARTK_OK;
    registry_remove(m);
    if (m->dtTmp) (void)hipFree(m->dtTmp);
    (void)hipHostFree(m->h_pin);
    (void)hipFree(m->arena);
    delete m;
    return DARTK_OK;
}

static int upload_pose_and_fk(Model *m, const float *h_T_mc, const float *h_joints, int mode, void *stream) {
    hipStream_t st = as_stream(stream);
    const int nJ = m->v.nJoints;
    PoseArgs p;
    memset(&p, 0, sizeof(p));
    if (h_T_mc) { p.hasT = 1; memcpy(p.T_mc, h_T_mc, 12 * sizeof(float)); }
    if (nJ > 0) {
        DARTK_REQUIRE(h_joints != nullptr, "joint array required for an articulated model");
        DARTK_REQUIRE(nJ <= DARTK_MAX_JOINTS, "%d joints exceed DARTK_MAX_JOINTS", nJ);
        p.nJ = nJ;
        memcpy(p.joints, h_joints, nJ * sizeof(float));
    }
    hipLaunchKernelGGL(fk_kernel, dim3(1), dim3(64), 0, st, m->v, mode, p);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_model_set_pose(dartk_model_t h, const float *h_T_mc, const float *h_joints, void *stream) {
    Model *m = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(m && h_T_mc, "dartk_model_set_pose: null argument");
    return upload_pose_and_fk(m, h_T_mc, h_joints, 0, stream);
}

int dartk_model_set_articulation(dartk_model_t h, const float *h_joints, void *stream) {
    Model *m = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(m, "dartk_model_set_articulation: null model");
    return upload_pose_and_fk(m, nullptr, h_joints, 1, stream);
}

int dartk_model_set_transforms(dartk_model_t h, const float *h_T_mc, const float *h_T_mf, const float *h_T_fm, void *stream) {
    Model *m = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(m && h_T_mc && h_T_mf && h_T_fm, "dartk_model_set_transforms: null argument");
    hipStream_t st = as_stream(stream);
    const int nF = m->v.nFrames;
    float T_cm[12];
    {   // SE3Invert on the host (pose.h keeps both directions)
        const float *A = h_T_mc;
        T_cm[0] = A[0]; T_cm[1] = A[4]; T_cm[2] = A[8];   T_cm[3]  = -(A[0] * A[3] + A[4] * A[7] + A[8] * A[11]);
        T_cm[4] = A[1]; T_cm[5] = A[5]; T_cm[6] = A[9];   T_cm[7]  = -(A[1] * A[3] + A[5] * A[7] + A[9] * A[11]);
        T_cm[8] = A[2]; T_cm[9] = A[6]; T_cm[10] = A[10]; T_cm[11] = -(A[2] * A[3] + A[6] * A[7] + A[10] * A[11]);
    }
    DARTK_HIP(hipMemcpyAsync(m->v.T_mc, h_T_mc, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipMemcpyAsync(m->v.T_cm, T_cm, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipMemcpyAsync(m->v.T_mf, h_T_mf, 12 * nF * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipMemcpyAsync(m->v.T_fm, h_T_fm, 12 * nF * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipStreamSynchronize(st));
    return DARTK_OK;
}

int dartk_model_get_transforms(dartk_model_t h, float *h_T_mc, float *h_T_mf, float *h_T_fm, float *h_joints) {
    Model *m = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(m, "dartk_model_get_transforms: null model");
    DARTK_HIP(hipDeviceSynchronize());
    const int nF = m->v.nFrames;
    if (h_T_mc) DARTK_HIP(hipMemcpy(h_T_mc, m->v.T_mc, 12 * sizeof(float), hipMemcpyDeviceToHost));
    if (h_T_mf) DARTK_HIP(hipMemcpy(h_T_mf, m->v.T_mf, 12 * nF * sizeof(float), hipMemcpyDeviceToHost));
    if (h_T_fm) DARTK_HIP(hipMemcpy(h_T_fm, m->v.T_fm, 12 * nF * sizeof(float), hipMemcpyDeviceToHost));
    if (h_joints && m->v.nJoints) DARTK_HIP(hipMemcpy(h_joints, m->v.joints, m->v.nJoints * sizeof(float), hipMemcpyDeviceToHost));
    return DARTK_OK;
}

// diagnostic library only (built with -DDARTK_COUNT_LOOKUPS): trilinear lookups performed since the last reset, by category
int dartk_lookup_counters(uint64_t *h_counts, int reset) {
    DARTK_REQUIRE(h_counts, "dartk_lookup_counters: null argument");
#ifdef DARTK_COUNT_LOOKUPS
    unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc;
    if ((rc = lookup_counters_obs2mod(tot, reset))) return rc;
    if ((rc = lookup_counters_gn(tot, reset))) return rc;
    if ((rc = lookup_counters_mod2obs(tot, reset))) return rc;
    if ((rc = lookup_counters_intersection(tot, reset))) return rc;
    if ((rc = lookup_counters_noobs(tot, reset))) return rc;
    for (int i = 0; i < 8; ++i) h_counts[i] = tot[i];
    return DARTK_OK;
#else
    (void)reset;
    set_error("dartk_lookup_counters: this library was built without lookup counters (load libdartk_count.so)");
    return DARTK_ERR_UNSUPPORTED;
#endif
}

int dartk_model_num_sdfs(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.nSdfs : 0; }
int dartk_model_dims(dartk_model_t h) { return h ? 6 + reinterpret_cast<Model *>(h)->v.nJoints : 0; }
float *dartk_model_obs_sdf_data(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.obsData : nullptr; }

}  // extern "C"
