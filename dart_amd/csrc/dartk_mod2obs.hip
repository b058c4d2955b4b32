// dartk_mod2obs.hip -- SDF ray-cast prediction, observation-SDF construction (splat + distance transform) and
// the model-to-observation normal equations.
// Replaces: optimization/kernels/raycast.cu:121-215,440-466; modToObs.cu:14-441,649-709 and wrappers :713-1064;
//           geometry/distance_transforms.cu:38-126,173-263,434-458,540-562; optimizer.cpp:168-219.
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_raycast.h"
#include "dartk_ne_bodies.h"

#include <cstdlib>

namespace dartk {

// -----------------------------------------------------------------------------------------------------
// ray-cast prediction (raycast.cu:121-215). The reference walks one ray per thread through every SDF in turn: a
// serial chain of ~nSdfs sphere-traces (each step a dependent 8-voxel gather) whose only coupling is the running
// closest hit. Here the (ray, SDF) traces are independent work items: wave = one 8x8 pixel tile against one SDF
// (tiles x nSdfs waves fill the chip; neighbouring rays march together through the same voxels). Each trace
// records the step position at which its first zero crossing was DETECTED and the refined hit; a second,
// per-pixel kernel replays the reference's sequential rule in SDF order -- a crossing counts iff its detection
// step lies before the closest hit so far (the reference's `t0 < closestT` loop test) -- so the map is
// identical to the serial walk; what is given up is the reference's pruning of traces behind the closest hit.
// -----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
raycast_trace_kernel(float flx, float fly, float ppx, float ppy, int width, int height, ModelView m_in,
                     float2 *__restrict__ trace /* [tasks][64], see stage_ray_rects */, float levelSet, FrameBatch fb) {
    __shared__ int sR[kRectInts];
    ModelView m = m_in;
    select_frame(m, blockIdx.y);                    // frame of a batch (0 otherwise)
    trace += (size_t)blockIdx.y * (fb.trace >> 1);
    stage_ray_rects(m, levelSet, flx, fly, ppx, ppy, width, height, sR);
    raycast_trace_tasks(blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, sR, flx, fly, ppx, ppy, width, height, m, trace, levelSet);
}

static constexpr int RC_REPLAY_SDFS = 64;    // per-SDF transforms staged in LDS, this many at a time
static constexpr int RC_REPLAY_GROUP = 8;    // trace records requested together by the replay

// the reference's sequential rule over SDFs [s0, s0+nc) for one pixel: a crossing counts iff it was detected before the
// closest hit so far
template <int GROUP = RC_REPLAY_GROUP>
__device__ __forceinline__ void replay_pixel(float4 &pr, float &closestT, int x, int y, int nSdfs, const int *sR, int modelNum, int s0, int nc,
                                             const float *sT, const float2 *__restrict__ trace, float flx, float fly, float ppx, float ppy) {
    const int tx = x >> 3, ty = y >> 3, sub = (y & 7) * 8 + (x & 7);
    // The rule is sequential in the SDF index, the trace records it reads are not: the records of GROUP SDFs are
    // requested together (one memory round trip instead of one per SDF whose rectangle holds the pixel), then judged in order.
    for (int k0 = 0; k0 < nc; k0 += GROUP) {
        float2 dh[GROUP];
#pragma unroll
        for (int g = 0; g < GROUP; ++g) {
            const int k = k0 + g, s = s0 + k;
            dh[g] = make_float2(3.0e38f, 0.0f);      // outside the SDF's rectangle nothing was traced: "no crossing"
            if (k < nc) {
                const int rx = tx - sR[s], ry = ty - sR[nSdfs + s], rw = sR[2 * nSdfs + s];
                const int off = sR[3 * nSdfs + s], cnt = sR[3 * nSdfs + s + 1] - off;
                if (!(rx < 0 || ry < 0 || rx >= rw || ry * rw + rx >= cnt)) dh[g] = trace[(size_t)(off + ry * rw + rx) * 64 + sub];
            }
        }
        // A pixel has a crossing with few of the SDFs, a wave of 64 pixels with most of them: each lane walks ITS crossings
        // (ascending SDF index, as the rule wants), so the wave takes as many turns as its busiest lane has crossings, not
        // one per SDF that any lane crosses.
        unsigned live = 0u;
#pragma unroll
        for (int g = 0; g < GROUP; ++g) live |= (dh[g].x < 3.0e38f) ? (1u << g) : 0u;
        while (live) {
            const int g = __builtin_ctz(live);
            live &= live - 1u;
            float det = dh[0].x, tHit = dh[0].y;
#pragma unroll
            for (int q = 1; q < GROUP; ++q) { det = (g == q) ? dh[q].x : det; tHit = (g == q) ? dh[q].y : tHit; }
            if (det < closestT) {
                const int k = k0 + g, s = s0 + k;
                const RayGeom gm = ray_in_frame(sT + 24 * k, x, y, flx, fly, ppx, ppy);
                const f3 xc = se3_apply_w(sT + 24 * k + 12, gm.dn.x * tHit + gm.of.x, gm.dn.y * tHit + gm.of.y, gm.dn.z * tHit + gm.of.z, 1.0f);
                if (pr.z == 0 || xc.z < pr.z) {
                    pr.x = xc.x; pr.y = xc.y; pr.z = xc.z; pr.w = (float)((modelNum << 16) + s);
                }
                closestT = tHit;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
raycast_replay_kernel(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, ModelView m,
                      const float2 *__restrict__ trace, float4 *__restrict__ prediction, float levelSet) {
    __shared__ float sT[RC_REPLAY_SDFS * 24];   // T_fc then T_cf per SDF
    __shared__ int sR[kRectInts];
    stage_ray_rects(m, levelSet, flx, fly, ppx, ppy, width, height, sR);
    const int nPix = width * height;
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = index < nPix;
    const int x = valid ? index % width : 0, y = valid ? index / width : 0;
    float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        pr = prediction[index];
        if (modelNum == 0) { pr.z = 0; pr.w = -1; }
    }
    float closestT = 10000;
    for (int s0 = 0; s0 < m.nSdfs; s0 += RC_REPLAY_SDFS) {
        const int nc = min(RC_REPLAY_SDFS, m.nSdfs - s0);
        __syncthreads();
        replay_stage_transforms(m, s0, nc, sT);
        __syncthreads();
        if (valid) replay_pixel(pr, closestT, x, y, m.nSdfs, sR, modelNum, s0, nc, sT, trace, flx, fly, ppx, ppy);
    }
    if (valid) prediction[index] = pr;
}

// every model's replay in model order plus cullUnobservable (modToObs.cu:682-709) in one launch: what the multi-model loop
// otherwise does with one replay launch per model and one cull launch, on its critical chain
struct ReplayMultiArgs {
    int n;
    ModelView m[DARTK_MAX_MODELS];
    unsigned long long off[DARTK_MAX_MODELS];       // each model's slice of the trace buffer (floats)
};
static_assert(sizeof(ReplayMultiArgs) <= 3584, "ReplayMultiArgs is passed by value as a kernel argument");

__global__ void __launch_bounds__(256)
raycast_replay_multi_kernel(float flx, float fly, float ppx, float ppy, int width, int height, ReplayMultiArgs a,
                            const float *__restrict__ rayBuf, float4 *__restrict__ prediction, const float4 *__restrict__ obs, int ow,
                            int oh, FrameBatch fb) {
    __shared__ float sT[RC_REPLAY_SDFS * 24];
    __shared__ int sR[kRectInts];
    const int frame = blockIdx.y;                   // frame of a batch (0 otherwise)
    rayBuf += (size_t)frame * fb.trace;
    prediction += (size_t)frame * fb.pred;
    obs += (size_t)frame * fb.vert;
    const int nPix = width * height;
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = index < nPix;
    const int x = valid ? index % width : 0, y = valid ? index / width : 0;
    float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) { pr = prediction[index]; pr.z = 0; pr.w = -1; }    // model 0 starts the map (x, y keep what the map held, as there)
    for (int mi = 0; mi < a.n; ++mi) {
        ModelView m = a.m[mi];
        select_frame(m, frame);
        const float2 *trace = reinterpret_cast<const float2 *>(rayBuf + a.off[mi]);
        float closestT = 10000;
        __syncthreads();                            // the previous model's table is no longer read
        stage_ray_rects(m, 0.0f, flx, fly, ppx, ppy, width, height, sR);
        for (int s0 = 0; s0 < m.nSdfs; s0 += RC_REPLAY_SDFS) {
            const int nc = min(RC_REPLAY_SDFS, m.nSdfs - s0);
            __syncthreads();
            replay_stage_transforms(m, s0, nc, sT);
            __syncthreads();
            if (valid) replay_pixel(pr, closestT, x, y, m.nSdfs, sR, mi, s0, nc, sT, trace, flx, fly, ppx, ppy);
        }
    }
    if (valid) {
        const int ox = x * ow / width, oy = y * oh / height;
        const int oi = ox + oy * ow;
        // cullUnobservable only ever clears z: pixels without a hit (most of the map) need not look at the observation
        if (pr.z != 0 && (obs[oi].w <= 0 || obs[oi + 1].w <= 0 || obs[oi + ow].w <= 0 || obs[oi + ow + 1].w <= 0)) pr.z = 0;
        prediction[index] = pr;
    }
}

// The same with every model's tables resident in LDS at once (dynamic LDS: per model the rectangle table [4 nSdfs + 1] and the
// per-SDF transforms [24 nSdfs]) and each workgroup walking MANY 256-pixel chunks of its frame. What a launch of the kernel
// above costs on a batch is not the pixels but its prologue -- per model a chain of dependent loads (SDF headers, pose,
// rectangles, transforms) and barriers, paid by every one of the 900 workgroups of a 640x360 frame: 650 us for 64 frames.
// Here a frame takes a few dozen workgroups and each pays the prologue once.
// tables (single frame only): the models' [transforms 24 n | rectangles 4 n + 1] of this pose, back to back in model order, as a
// UK_TABLES workgroup of the launch before left them (replay_tables_to_global) -- the prologue is then one copy, no chain of
// dependent loads per model. keepXY: x, y of a pixel without a hit stay what the caller's map held (the reference's behaviour).
__global__ void __launch_bounds__(256)
raycast_replay_multi_rows_kernel(float flx, float fly, float ppx, float ppy, int width, int height, ReplayMultiArgs a,
                                 const float *__restrict__ rayBuf, float4 *__restrict__ prediction, const float4 *__restrict__ obs, int ow,
                                 int oh, FrameBatch fb, int *__restrict__ hitCount, const float *__restrict__ tables, int keepXY) {
    extern __shared__ float lds[];
    const int frame = blockIdx.y;
    rayBuf += (size_t)frame * fb.trace;
    prediction += (size_t)frame * fb.pred;
    obs += (size_t)frame * fb.vert;
    const int nPix = width * height;
    if (tables) {
        int off = 0;
        for (int mi = 0; mi < a.n; ++mi) {
            const int n = a.m[mi].nSdfs, nR = 4 * n + 1;
            for (int i = threadIdx.x; i < 28 * n + 1; i += blockDim.x) lds[off + i] = (i < nR) ? tables[off + 24 * n + i] : tables[off + i - nR];
            off += 28 * n + 1;
        }
        __syncthreads();
    } else {
        int off = 0;
        for (int mi = 0; mi < a.n; ++mi) {
            ModelView m = a.m[mi];
            select_frame(m, frame);
            const int n = m.nSdfs;
            stage_ray_rects(m, 0.0f, flx, fly, ppx, ppy, width, height, reinterpret_cast<int *>(lds + off));
            replay_stage_transforms(m, 0, n, lds + off + 4 * n + 1);
            off += 28 * n + 1;
        }
        __syncthreads();
    }
    // the tile rectangle that holds every SDF's rectangle of every model: pixels outside it (most of a frame) cannot have a
    // trace record, and a wave whose 64 pixels all lie outside skips the per-SDF walk altogether
    __shared__ int sUnion[4];
    if (threadIdx.x == 0) {
        int x0 = 1 << 30, y0 = 1 << 30, x1 = -1, y1 = -1, off = 0;
        for (int mi = 0; mi < a.n; ++mi) {
            const int n = a.m[mi].nSdfs;
            const int *sR = reinterpret_cast<const int *>(lds + off);
            for (int s = 0; s < n; ++s) {
                const int rw = sR[2 * n + s], cnt = sR[3 * n + s + 1] - sR[3 * n + s];
                if (rw <= 0 || cnt <= 0) continue;
                x0 = min(x0, sR[s]); y0 = min(y0, sR[n + s]);
                x1 = max(x1, sR[s] + rw - 1); y1 = max(y1, sR[n + s] + (cnt + rw - 1) / rw - 1);
            }
            off += 28 * n + 1;
        }
        sUnion[0] = x0; sUnion[1] = y0; sUnion[2] = x1; sUnion[3] = y1;
    }
    __syncthreads();
    const int ux0 = sUnion[0], uy0 = sUnion[1], ux1 = sUnion[2], uy1 = sUnion[3];
    const int nChunks = (nPix + 255) / 256;
    if (hitCount) hitCount += (size_t)frame * fb.counts;
    for (int chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
        const int index = chunk * 256 + threadIdx.x;
        const bool inMap = index < nPix;
        const int x = inMap ? index % width : 0, y = inMap ? index / width : 0;
        // model 0 starts the map. The reference leaves x, y of a pixel without a hit as they were; nothing reads them (every
        // consumer tests z first), and this kernel only ever writes the loop's own scratch map, so they are not fetched
        float4 pr = make_float4(0.f, 0.f, 0.f, -1.f);
        if (keepXY && inMap) { pr = prediction[index]; pr.z = 0.f; pr.w = -1.f; }
        if (inMap && (x >> 3) >= ux0 && (x >> 3) <= ux1 && (y >> 3) >= uy0 && (y >> 3) <= uy1) {
            int off = 0;
            for (int mi = 0; mi < a.n; ++mi) {
                const int n = a.m[mi].nSdfs;
                float closestT = 10000;
                replay_pixel(pr, closestT, x, y, n, reinterpret_cast<const int *>(lds + off), mi, 0, n, lds + off + 4 * n + 1,
                             reinterpret_cast<const float2 *>(rayBuf + a.off[mi]), flx, fly, ppx, ppy);
                off += 28 * n + 1;
            }
            const int ox = x * ow / width, oy = y * oh / height;
            const int oi = ox + oy * ow;
            if (pr.z != 0 && (obs[oi].w <= 0 || obs[oi + 1].w <= 0 || obs[oi + ow].w <= 0 || obs[oi + ow + 1].w <= 0)) pr.z = 0;
        }
        if (!hitCount) {
            if (inMap) prediction[index] = pr;
            continue;
        }
        // hit-list form: the frame's map region holds its hits back to back instead (a few percent of the pixels), appended per
        // wave; the consumers walk hitCount entries. Neither the pixels without a hit nor their positions are written.
        const bool hit = inMap && pr.z != 0;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(hitCount, __popcll(mask));
            base = __shfl(base, 0);
            if (hit) prediction[base + __popcll(mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = pr;
        }
    }
}

// the trace alone (the replay is done by the consumer: ne_mod2obs_kernel<.., REPLAY = true>)
static int launch_raycast_trace(float flx, float fly, float ppx, float ppy, int width, int height, const ModelView &m, float levelSet,
                                Workspace *ws, hipStream_t st, size_t bufOffset = 0) {
    const size_t nPix = (size_t)width * height;
    (void)nPix;
    int rc = cur_batch().n > 1 ? DARTK_OK : ensure_ray_buf(ws, bufOffset + ray_trace_floats(width, height, m.nSdfs));     // (callers with an offset / a batch have sized the buffer already)
    if (rc) return rc;
    const long long tasks = (long long)((width + 7) / 8) * ((height + 7) / 8) * m.nSdfs;
    DARTK_REQUIRE(tasks < (1ll << 31), "ray-cast of %dx%d against %d SDFs exceeds the launch limits", width, height, m.nSdfs);
    DARTK_REQUIRE(m.nSdfs <= kMaxRaySdfs, "ray-cast supports at most %d SDFs per model, got %d", kMaxRaySdfs, m.nSdfs);
    // the tasks that survive the tile rectangles are known on the device only: a fixed grid walks them (4 waves per workgroup,
    // one task per wave and pass); a single frame gets enough workgroups for the whole chip, a batch shares it among its frames
    // (measured sweep on 64 frames, tools/time_batch_replay.sh: 24-96 per frame equal, 192 +2 %, 384 +7 %)
    static const char *envT = getenv("DARTK_TRACE_WGS");      // diagnostic override: workgroups per frame of a batch
    const long long want = cur_batch().n > 1 ? ((envT && atoi(envT) > 0) ? atoi(envT) : 64) : 1024;
    const unsigned blocks = (unsigned)((tasks + 3) / 4 < want ? (tasks + 3) / 4 : want);
    ProfScope ps(DARTK_PROF_RAYCAST, st);
    hipLaunchKernelGGL(raycast_trace_kernel, dim3(blocks, cur_batch().n), dim3(256), 0, st, flx, fly, ppx, ppy, width, height, m,
                       (float2 *)(ws->rayBuf + bufOffset), levelSet, cur_batch());
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

static int launch_raycast(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                          float *d_prediction, float levelSet, hipStream_t st) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    rc = launch_raycast_trace(flx, fly, ppx, ppy, width, height, m, levelSet, ws, st);
    if (rc) return rc;
    const size_t nPix = (size_t)width * height;
    ProfScope ps2(DARTK_PROF_RAYCAST_REPLAY, st);
    hipLaunchKernelGGL(raycast_replay_kernel, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, st, flx, fly, ppx, ppy, width, height,
                       modelNum, m, (const float2 *)ws->rayBuf, (float4 *)d_prediction, levelSet);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// gpu_cullUnobservable, modToObs.cu:682-709
__global__ void cull_unobservable_kernel(float4 *__restrict__ pred, int pw, int ph, const float4 *__restrict__ obs, int ow, int oh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pw * ph) return;
    const int px = i % pw, py = i / pw;
    const int ox = px * ow / pw, oy = py * oh / ph;
    const int oi = ox + oy * ow;
    if (obs[oi].w <= 0 || obs[oi + 1].w <= 0 || obs[oi + ow].w <= 0 || obs[oi + ow + 1].w <= 0) pred[i].z = 0;
}

// gpu_splatObsSdf, modToObs.cu:403-441
// dst: where the splat goes ([frames][voxels]; the distance transform that follows ends in the model's obsData).
// T_gc_out (optional, single frame): receives the camera->model transform the splat was made with (the observation SDF's frame).
__global__ void splat_obs_sdf_kernel(const float4 *__restrict__ vert, int width, int height, ModelView m_in, float focalLength, FrameBatch fb,
                                     float *__restrict__ dst, float *__restrict__ T_gc_out) {
    ModelView m = m_in;
    select_frame(m, blockIdx.y);                    // frame of a batch (0 otherwise)
    vert += (size_t)blockIdx.y * fb.vert;
    const int nVox = m.obs.dimx * m.obs.dimy * m.obs.dimz;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (T_gc_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 12) T_gc_out[threadIdx.x] = m.T_mc[threadIdx.x];
    if (i >= nVox) return;
    const int x = i % m.obs.dimx, y = (i / m.obs.dimx) % m.obs.dimy, z = i / (m.obs.dimx * m.obs.dimy);
    const float *__restrict__ T_cm = m.T_cm;
    f3 p; p.x = m.obs.ox + m.obs.res * (float)x; p.y = m.obs.oy + m.obs.res * (float)y; p.z = m.obs.oz + m.obs.res * (float)z;
    const f3 c = se3_apply(T_cm, p);
    const int u = (int)roundf((focalLength / c.z) * c.x + (float)(width >> 1));
    const int v = (int)roundf((focalLength / c.z) * c.y + (float)(height >> 1));
    float val;
    if (u < 0 || u >= width || v < 0 || v >= height) val = 1000.0f;
    else {
        const float4 o = vert[u + v * width];
        if (o.w == 0 || o.z == 0) val = 500.0f;
        else if (o.z < c.z) val = 0.0f;
        else val = 1000.0f;
    }
    dst[(size_t)blockIdx.y * nVox + i] = val;
}

// -----------------------------------------------------------------------------------------------------
// distance transform. The reference runs Felzenszwalb's serial lower-envelope scan with one thread per grid
// line (distance_transforms.cu:38-126). On a 64-lane machine with 160 KB of LDS the direct min-plus form
//     out[q] = min_p ( (q-p)^2 + in[p] )
// is both parallel over q and cheap (len <= 512): one lane per output voxel, the line resident in LDS. For the
// integer-valued inputs these transforms see (0 / 500 / 1000 / 1e20 and squared integer distances, all exact
// in fp32) the minimum is exact, i.e. identical to what the envelope scan returns.
// -----------------------------------------------------------------------------------------------------
// along x (stride 1): one workgroup per group of lines
__global__ void __launch_bounds__(256)
dt_pass_x_kernel(const float *__restrict__ in, float *__restrict__ out, int len, int nLines, int takeSqrt) {
    extern __shared__ float lds[];
    const int linesPerBlock = max(1, 256 / len);
    const int line0 = blockIdx.x * linesPerBlock;
    const int nl = min(linesPerBlock, nLines - line0);
    const int total = nl * len;
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = in[(size_t)line0 * len + i];
    __syncthreads();
    // (q - p)^2 + in[p] as ONE fma of the float difference: the square of an integer below 2^12 is exact, so the fma rounds the
    // same sum once, as the multiply-then-add did -- two operations less per (q, p) pair
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int l = i / len, q = i - l * len;
        const float *row = lds + l * len;
        float best = 3.0e38f;
        float d = (float)q;
#pragma unroll 8
        for (int p = 0; p < len; ++p) {
            best = fminf(best, __builtin_fmaf(d, d, row[p]));
            d -= 1.0f;
        }
        out[(size_t)line0 * len + i] = takeSqrt ? sqrtf(best) : best;
    }
}
// along a strided axis: workgroup = 64 consecutive x by the whole line in LDS, producing DT_QCHUNK outputs of it;
// lane (x, group) computes 4 outputs that share every LDS read. grid = tilesX * nOuter * ceil(len/DT_QCHUNK).
static constexpr int DT_QCHUNK = 16;
__global__ void __launch_bounds__(256)
dt_pass_strided_kernel(const float *__restrict__ in, float *__restrict__ out, int len, int stride, int width, int nOuter,
                       int outerStride, int takeSqrt, size_t volumeStride) {
    extern __shared__ float lds[];      // [len][64]
    in += (size_t)blockIdx.y * volumeStride;        // volume blockIdx.y of a batch of equally sized volumes
    out += (size_t)blockIdx.y * volumeStride;
    const int tilesX = (width + 63) / 64;
    const int qChunks = (len + DT_QCHUNK - 1) / DT_QCHUNK;
    const int qc = blockIdx.x % qChunks;
    const int rest = blockIdx.x / qChunks;
    const int x0 = (rest % tilesX) * 64;
    const int outer = rest / tilesX;
    if (outer >= nOuter) return;
    const int lx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t base = (size_t)outer * outerStride + x0 + lx;
    const bool ok = (x0 + lx) < width;
    for (int p = g; p < len; p += 4) lds[p * 64 + lx] = ok ? in[base + (size_t)p * stride] : 3.0e38f;
    __syncthreads();
    if (!ok) return;
    const int q0 = qc * DT_QCHUNK + g * 4;
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f, b3 = 3.0e38f;
    float d0 = (float)q0, d1 = (float)(q0 + 1), d2 = (float)(q0 + 2), d3 = (float)(q0 + 3);     // q - p as floats, one fma per pair (see above)
#pragma unroll 8
    for (int p = 0; p < len; ++p) {
        const float v = lds[p * 64 + lx];
        b0 = fminf(b0, __builtin_fmaf(d0, d0, v));
        b1 = fminf(b1, __builtin_fmaf(d1, d1, v));
        b2 = fminf(b2, __builtin_fmaf(d2, d2, v));
        b3 = fminf(b3, __builtin_fmaf(d3, d3, v));
        d0 -= 1.0f; d1 -= 1.0f; d2 -= 1.0f; d3 -= 1.0f;
    }
    const float r[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q0 + k < len) out[base + (size_t)(q0 + k) * stride] = takeSqrt ? sqrtf(r[k]) : r[k];
}

// nVolumes equally sized volumes, back to back, are transformed by the same three launches
int launch_distance_transform_3d(float *in, float *out, int width, int height, int depth, int takeSqrt, hipStream_t st, int nVolumes = 1) {
    DARTK_REQUIRE(width >= 1 && height >= 1 && depth >= 1 && width <= 512 && height <= 512 && depth <= 512,
                  "distance transform dimensions must be in [1,512], got %dx%dx%d", width, height, depth);
    const size_t volume = (size_t)width * height * depth;
    // x: in -> out ; z: out -> in ; y: in -> out (sqrt)   (distance_transforms.cu:440-456)
    {
        const int nLines = height * depth * nVolumes;
        const int lpb = 256 / width > 0 ? 256 / width : 1;
        hipLaunchKernelGGL(dt_pass_x_kernel, dim3((nLines + lpb - 1) / lpb), dim3(256), sizeof(float) * (size_t)(lpb * width), st,
                           in, out, width, nLines, 0);
    }
    const int tilesX = (width + 63) / 64;
    const int qcZ = (depth + DT_QCHUNK - 1) / DT_QCHUNK, qcY = (height + DT_QCHUNK - 1) / DT_QCHUNK;
    hipLaunchKernelGGL(dt_pass_strided_kernel, dim3(tilesX * height * qcZ, nVolumes), dim3(256), sizeof(float) * 64 * (size_t)depth, st,
                       out, in, depth, width * height, width, height, width, 0, volume);
    hipLaunchKernelGGL(dt_pass_strided_kernel, dim3(tilesX * depth * qcY, nVolumes), dim3(256), sizeof(float) * 64 * (size_t)height, st,
                       in, out, height, width, width, depth, width * height, takeSqrt, volume);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// gpu_seedInverseTransform3D (distance_transforms.cu:173-227) / gpu_combineTransforms3D (:248-263)
__global__ void dt_seed_kernel(const float *__restrict__ im, float *__restrict__ seed, int w, int h, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h * d) return;
    const int x = i % w, y = (i / w) % h, z = i / (w * h);
    float s = 0.0f;
    if (im[i] == 0) {
        s = 1e20f;
        if ((x > 0 && im[i - 1] != 0.0f) || (x < w - 1 && im[i + 1] != 0.0f) || (y > 0 && im[i - w] != 0.0f) ||
            (y < h - 1 && im[i + w] != 0.0f) || (z > 0 && im[i - w * h] != 0.0f) || (z < d - 1 && im[i + w * h] != 0.0f))
            s = 0.0f;
    }
    seed[i] = s;
}
__global__ void dt_combine_kernel(float *__restrict__ im, const float *__restrict__ neg, const float *__restrict__ mask, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i] == 0) im[i] = -neg[i];
}

// -----------------------------------------------------------------------------------------------------
// mod -> obs normal equations (modToObs.cu:14-401). One lane per prediction pixel; pixels that contribute
// are compacted (deterministically) into LDS rows before the pair reduction, so empty image regions cost one
// ballot. RED: 0 full, 1 linear, 2 param map. TRUNC: truncation window (modToObs.cu:163-166).
// -----------------------------------------------------------------------------------------------------
#ifdef DARTK_STAMPS
__device__ unsigned long long g_m2oStamps[1024 * 8];     // per workgroup of the last launch: realtime ticks (10 ns) at the phase boundaries
#define M2O_STAMP(i) do { stp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define M2O_STAMP(i) do { } while (0)
#endif
// a batch's replay tables: one workgroup per frame leaves what replay_tables_to_global computes for the frame's pose
__global__ void __launch_bounds__(NE_BLOCK)
replay_tables_kernel(ModelView m_in, float flx, float fly, float ppx, float ppy, int width, int height, float *__restrict__ tables, int stride) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    select_frame(m, blockIdx.x);
    replay_tables_to_global(m, flx, fly, ppx, ppy, width, height, lds, tables + (size_t)blockIdx.x * stride);
}

template <int PPT, int RED, bool TRUNC, int REPLAY = 0>       // REPLAY: 0 reads the map, 1 forms it from its model's traces, 2 from every model's
__device__ __forceinline__ void ne_mod2obs_body(const float4 *__restrict__ pred, int nPix, int modelNum, ModelView m_in, const float *__restrict__ T_gc_in,
                  float truncationDist, int redDims, const float *__restrict__ dtheta_dalpha, const int *__restrict__ mapping,
                  float *__restrict__ slabs, int *__restrict__ numPredictions, ReplayArgs ra, FrameBatch fb, int tgcStride,
                  const int *__restrict__ nItems, int bx, int gx, int by) {
    extern __shared__ float lds[];
#ifdef DARTK_STAMPS
    unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    M2O_STAMP(0);
    ModelView m = m_in;
    select_frame(m, by);                    // frame of a batch (0 otherwise)
    pred += (size_t)by * fb.pred;
    T_gc_in += (size_t)by * tgcStride;
    numPredictions += (size_t)by * fb.counts;
    if (REPLAY) {       // the frame's slice of the trace buffer (fb.trace floats apart) and of the prediction maps
        ra.trace += (size_t)by * (fb.trace / 2);
        if (ra.predOut) ra.predOut += (size_t)by * fb.pred;
        if (ra.tables) ra.tables += (size_t)by * ra.tablesStride;
    }
    // hit-list form (batches): `pred` holds the frame's hits back to back, nItems[frame] of them (raycast_replay_multi_rows_kernel)
    if (nItems) nPix = nItems[(size_t)by * fb.counts];
    const int fullDims = 6 + m.nJoints;
    const int d = RED ? redDims : fullDims;
    const int ld = row_len(d, false);
    float *sT_fm = lds;
    float *sT_mf = sT_fm + 12 * m.nFrames;
    const int fullLd = fullDims | 1;
    float *sFull = sT_mf + 12 * m.nFrames;
    float *rows = sFull + (RED ? NE_BLOCK * fullLd : 0);
    int *waveCnt = reinterpret_cast<int *>(rows + NE_BLOCK * ld);
    unsigned *sDep = reinterpret_cast<unsigned *>(waveCnt + 8);    // each frame's dependency mask (low, high word)
    int *sFrameOf = reinterpret_cast<int *>(sDep + 2 * m.nFrames);  // each SDF's frame
    int *sJType = sFrameOf + m.nSdfs;                               // joint types and axes, indexed per lane by the Jacobian walk
    float *sJAxis = reinterpret_cast<float *>(sJType + m.nJoints);
    float *sRT = sJAxis + 3 * m.nJoints;                            // REPLAY: per-SDF ray transforms, 24 * nSdfs floats
    int *sR = reinterpret_cast<int *>(sRT + 24 * m.nSdfs);     // REPLAY: the tile rectangles of the sparse traces, 4 * nSdfs + 1 ints
    // Everything the workgroup stages is requested before anything is waited for: the frame transforms, the replay tables and
    // the first tile's prediction pixel travel together (one memory round trip; a loop of load -> LDS store pays one per turn)
    constexpr int QS = 4;
    int nTabAll = 28 * m.nSdfs + 1;
    if (REPLAY == 2) { nTabAll = 0; for (int mi = 0; mi < ra.multi->nModels; ++mi) nTabAll += 28 * ra.multi->nSdfsOf[mi] + 1; }
    const int nT = 12 * m.nFrames, nTab = (REPLAY && ra.tables) ? nTabAll : 0;
    const bool batched = nT <= QS * NE_BLOCK && nTab <= QS * NE_BLOCK;
    float4 pcFirst = make_float4(0.f, 0.f, 0.f, 0.f);
    if (REPLAY && ra.predOut && bx * NE_BLOCK + (int)threadIdx.x < nPix) pcFirst = ra.predOut[bx * NE_BLOCK + threadIdx.x];
    // the label -> frame -> dependency-mask chain of a contributing pixel is served from LDS
    for (int i = threadIdx.x; i < 2 * m.nFrames + m.nSdfs; i += NE_BLOCK)
        sDep[i] = i < 2 * m.nFrames ? reinterpret_cast<const unsigned *>(m.depMask)[i] : (unsigned)m.sdfFrame[i - 2 * m.nFrames];
    for (int i = threadIdx.x; i < 4 * m.nJoints; i += NE_BLOCK) {
        if (i < m.nJoints) sJType[i] = m.jointType[i];
        else sJAxis[i - m.nJoints] = m.jointAxis[i - m.nJoints];
    }
    if (batched) {
        float vf[QS], vm[QS], vt[QS];
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            const int i = threadIdx.x + q * NE_BLOCK;
            vf[q] = i < nT ? m.T_fm[i] : 0.0f;
            vm[q] = i < nT ? m.T_mf[i] : 0.0f;
            vt[q] = i < nTab ? ra.tables[i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            const int i = threadIdx.x + q * NE_BLOCK;
            if (i < nT) { sT_fm[i] = vf[q]; sT_mf[i] = vm[q]; }
            if (i < nTab) sRT[i] = vt[q];                       // (sR follows sRT in LDS as in the table)
        }
    } else {
        stage_model(m, sT_fm, sT_mf, nullptr);
        for (int i = threadIdx.x; i < nTab; i += blockDim.x) sRT[i] = ra.tables[i];
    }
    if (REPLAY) {       // nSdfs <= NE_BLOCK checked by the launcher; every thread reads sRT in the first tile already
        if (ra.tables) {
            __syncthreads();
        } else {
            replay_stage_transforms(m, 0, m.nSdfs, sRT);
            stage_ray_rects(m, 0.0f, ra.flx, ra.fly, ra.ppx, ra.ppy, ra.width, nPix / ra.width, sR);
        }
    }
    float T_gc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T_gc[i] = T_gc_in[i];
    const SdfDev obs = m.obs;
    const float *__restrict__ obsData = m.obsData;

    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, false);
    int myCount = 0;
    M2O_STAMP(1);
    const int nTiles = (nPix + NE_BLOCK - 1) / NE_BLOCK;
    for (int tile = bx; tile < nTiles; tile += gx) {
        const int index = tile * NE_BLOCK + threadIdx.x;
        bool active = false;
        f3 pm, pg;
        float residual = 0.0f;
        int frame = 0;
        if (index < nPix) {
            float4 pc;
            if (REPLAY == 2) {
                const ReplayMulti &mu = *ra.multi;
                pc = (tile == (int)bx || !ra.predOut) ? pcFirst : ra.predOut[index];
                pc.z = 0; pc.w = -1;                     // model 0 starts the map (x, y keep what the map held)
                const int px = index % ra.width, py = index / ra.width;
                int off = 0;
                for (int mi = 0; mi < mu.nModels; ++mi) {
                    const int n = mu.nSdfsOf[mi];
                    float closestT = 10000;
                    replay_pixel<24>(pc, closestT, px, py, n, reinterpret_cast<const int *>(sRT + off + 24 * n), mi, 0, n, sRT + off,
                                     reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(ra.trace) + mu.traceOff[mi]),
                                     ra.flx, ra.fly, ra.ppx, ra.ppy);
                    off += 28 * n + 1;
                }
                if (pc.z != 0) {                         // cullUnobservable only ever clears z
                    const int height = nPix / ra.width;
                    const int ox = px * mu.ow / ra.width, oy = py * mu.oh / height;
                    const int oi = ox + oy * mu.ow;
                    if (mu.obs[oi].w <= 0 || mu.obs[oi + 1].w <= 0 || mu.obs[oi + mu.ow].w <= 0 || mu.obs[oi + mu.ow + 1].w <= 0) pc.z = 0;
                }
                if (mu.writesMap && ra.predOut) ra.predOut[index] = pc;
            } else if (REPLAY) {       // the pixel's prediction is formed here from the per-SDF traces and kept for the caller
                // (no map: model 0 of a loop that returns no prediction -- the replay starts every pixel from "no hit" anyway)
                pc = (tile == (int)bx || !ra.predOut) ? pcFirst : ra.predOut[index];
                M2O_STAMP(2);
                if (ra.modelNum == 0) { pc.z = 0; pc.w = -1; }
                float closestT = 10000;
                replay_pixel<24>(pc, closestT, index % ra.width, index / ra.width, m.nSdfs, sR, ra.modelNum, 0, m.nSdfs, sRT, ra.trace,
                             ra.flx, ra.fly, ra.ppx, ra.ppy);
                M2O_STAMP(3);
                if (ra.predOut) ra.predOut[index] = pc;
            } else {
                pc = pred[index];
            }
            M2O_STAMP(4);
            if (pc.z != 0) {
                f3 p3; p3.x = pc.x; p3.y = pc.y; p3.z = pc.z;
                pm = se3_apply(T_gc, p3);
                pg = grid_coords(obs, pm);
                if (in_bounds_grad_interp(obs, pg)) {
                    residual = grid_value<CNT_M2O>(obsData, obs.dimx, obs.dimy, pg) * obs.res;
                    bool ok = true;
                    if (TRUNC) ok = !(residual >= truncationDist || residual < 0);
                    if (ok) {
                        const int label = (int)roundf(pc.w);
                        if ((label >> 16) == modelNum) {
                            frame = sFrameOf[label & 65535];
                            active = true;
                        }
                    }
                }
            }
        }
        M2O_STAMP(5);
        int nActive;
        const int slot = compact_slot<NE_BLOCK>(active, waveCnt, nActive);
        if (active) {
            float *row = rows + slot * ld;
            const f3 gm = grid_gradient<CNT_M2O>(obsData, obs.dimx, obs.dimy, pg);
            float *Jfull = RED ? (sFull + threadIdx.x * fullLd) : row;
            const unsigned long long dep = (unsigned long long)sDep[2 * frame] | ((unsigned long long)sDep[2 * frame + 1] << 32);
            error_jacobian_walk<false>(Jfull, pm, 1.0f, dep, gm, m.nJoints, sJType, sJAxis, sT_fm, sT_mf);
            if (RED) reduce_jacobian<RED>(row, Jfull, 6, fullDims, redDims, dtheta_dalpha, mapping);
            row[d] = residual;
            row[d + 1] = (float)(0.5 * (double)residual * (double)residual);
            row[d + 2] = 1.0f;
        }
        __syncthreads();
        M2O_STAMP(6);
        if (nActive) acc.accumulate(rows, ld, nActive);
        myCount += nActive;
        __syncthreads();
        if (REPLAY && ra.slabPerTile) {
            acc.store(slabs + ((size_t)by * nTiles + tile) * sys_size(d), sys_size(d));
            acc.init(d, false);
        }
    }
    if (!(REPLAY && ra.slabPerTile)) acc.store(slabs + ((size_t)by * gx + bx) * sys_size(d), sys_size(d));
    if (threadIdx.x == 0 && myCount) atomicAdd(numPredictions, myCount);
#ifdef DARTK_STAMPS
    M2O_STAMP(7);
    if (REPLAY && threadIdx.x == 0 && bx < 1024)
        for (int i = 0; i < 8; ++i) g_m2oStamps[8 * bx + i] = stp[i];
#endif
}

template <int PPT, int RED, bool TRUNC, int REPLAY = 0>
__global__ void __launch_bounds__(NE_BLOCK)
ne_mod2obs_kernel(const float4 *__restrict__ pred, int nPix, int modelNum, ModelView m_in, const float *__restrict__ T_gc_in,
                  float truncationDist, int redDims, const float *__restrict__ dtheta_dalpha, const int *__restrict__ mapping,
                  float *__restrict__ slabs, int *__restrict__ numPredictions, ReplayArgs ra, FrameBatch fb, int tgcStride,
                  const int *__restrict__ nItems) {
    ne_mod2obs_body<PPT, RED, TRUNC, REPLAY>(pred, nPix, modelNum, m_in, T_gc_in, truncationDist, redDims, dtheta_dalpha, mapping, slabs, numPredictions, ra, fb, tgcStride, nItems, blockIdx.x, gridDim.x, blockIdx.y);
}


// -----------------------------------------------------------------------------------------------------
// Several term kernels in one launch (UnionTable, dartk_kernels.h): the workgroup finds its segment and runs that segment's body
// with its virtual index. Bodies of different terms never share a workgroup, so LDS and barriers are each body's own.
// -----------------------------------------------------------------------------------------------------
// GROUP 0: trace / association / intersection bodies, GROUP 1: mod->obs / obs->mod bodies -- two kernels, each with the register
// budget of its own bodies. (Measured: with the intersection bodies, 150 registers, moved to group 1 the first launch is no shorter
// -- it lasts as long as the trace's stragglers, 40 us at 3 or at 8 waves per SIMD -- and the second grows from 26 to 33 us.)
template <int PPT, int GROUP>
__global__ void __launch_bounds__(NE_BLOCK)
union_terms_kernel(const UnionTable *__restrict__ t) {
    extern __shared__ float lds[];
    int s = 0;
    const int n = t->nSeg;
    while (s + 1 < n && (int)blockIdx.x >= t->seg[s + 1].first) ++s;
    const UnionSeg &g = t->seg[s];
    const int bx = (int)blockIdx.x - g.first, gx = g.count;
    const FrameBatch one{1, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (GROUP == 0) {
    switch (g.kind) {
    case UK_TRACE: {
        int *sR = reinterpret_cast<int *>(lds);
        stage_ray_rects(g.m, 0.0f, g.flx, g.fly, g.ppx, g.ppy, g.width, g.height, sR);
        constexpr int W = NE_BLOCK / 64;
        raycast_trace_tasks(bx * W + (int)(threadIdx.x >> 6), gx * W, sR, g.flx, g.fly, g.ppx, g.ppy, g.width, g.height, g.m, g.trace, 0.0f);
        break;
    }
    case UK_ASSOC:
        assoc_multi_body<false, true>(g.vert, g.nPix, g.assocArgs, g.lastElement, nullptr, nullptr, one, g.validIdx, g.nValid, bx, gx, 0);
        break;
    case UK_SELF:
        if (g.red == 0) ne_self_intersection_body<PPT, 0>(g.sites, g.nSites, g.m, g.pot, g.redDims, g.A, g.map, g.slabs, bx, gx, 0);
        else if (g.red == 1) ne_self_intersection_body<PPT, 1>(g.sites, g.nSites, g.m, g.pot, g.redDims, g.A, g.map, g.slabs, bx, gx, 0);
        else ne_self_intersection_body<PPT, 2>(g.sites, g.nSites, g.m, g.pot, g.redDims, g.A, g.map, g.slabs, bx, gx, 0);
        break;
    case UK_INTER:
        if (g.red == 0) ne_intersection_body<PPT, 0>(g.sites, g.nSites, g.m, g.m2, nullptr, g.redDims, g.A, g.map, g.slabs, 0, bx, gx, 0);
        else if (g.red == 1) ne_intersection_body<PPT, 1>(g.sites, g.nSites, g.m, g.m2, nullptr, g.redDims, g.A, g.map, g.slabs, 0, bx, gx, 0);
        else ne_intersection_body<PPT, 2>(g.sites, g.nSites, g.m, g.m2, nullptr, g.redDims, g.A, g.map, g.slabs, 0, bx, gx, 0);
        break;
    case UK_TABLES:
        replay_tables_to_global(g.m, g.flx, g.fly, g.ppx, g.ppy, g.width, g.height, lds, g.slabs);
        break;
    default: break;
    }
    } else {
    switch (g.kind) {
    case UK_M2O:
        if (g.replay.tables) {      // the replay of every model's traces and the cull happen in the term itself: no launch between
            if (g.red == 0) ne_mod2obs_body<PPT, 0, false, 2>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, g.replay, one, 0, nullptr, bx, gx, 0);
            else if (g.red == 1) ne_mod2obs_body<PPT, 1, false, 2>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, g.replay, one, 0, nullptr, bx, gx, 0);
            else ne_mod2obs_body<PPT, 2, false, 2>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, g.replay, one, 0, nullptr, bx, gx, 0);
            break;
        }
        if (g.red == 0) ne_mod2obs_body<PPT, 0, false, 0>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, ReplayArgs{}, one, 0, nullptr, bx, gx, 0);
        else if (g.red == 1) ne_mod2obs_body<PPT, 1, false, 0>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, ReplayArgs{}, one, 0, nullptr, bx, gx, 0);
        else ne_mod2obs_body<PPT, 2, false, 0>(g.pred, g.nPred, g.m.modelId, g.m, g.T_gc, 0.f, g.redDims, g.A, g.map, g.slabs, g.numPredictions, ReplayArgs{}, one, 0, nullptr, bx, gx, 0);
        break;
    default:        // UK_O2M
        if (g.red == 0) ne_obs2mod_body<PPT, 0>(g.pts, g.vert, g.nPix, g.m, g.huberDelta, g.redDims, g.A, g.map, g.slabs, g.count_, one, bx, gx, 0);
        else if (g.red == 1) ne_obs2mod_body<PPT, 1>(g.pts, g.vert, g.nPix, g.m, g.huberDelta, g.redDims, g.A, g.map, g.slabs, g.count_, one, bx, gx, 0);
        else ne_obs2mod_body<PPT, 2>(g.pts, g.vert, g.nPix, g.m, g.huberDelta, g.redDims, g.A, g.map, g.slabs, g.count_, one, bx, gx, 0);
        break;
    }
    }
}

template <int PPT, int GROUP>
static int launch_union(const UnionTable *d_table, int totalBlocks, size_t ldsBytes, hipStream_t st) {
    auto kfn = union_terms_kernel<PPT, GROUP>;
    if (ldsBytes > 64 * 1024) DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kfn, dim3(totalBlocks), dim3(NE_BLOCK), ldsBytes, st, d_table);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
int gn_launch_union(const UnionTable *d_table, int totalBlocks, int ppt, int group, size_t ldsBytes, hipStream_t st) {
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
    DARTK_REQUIRE((ppt == 1 || ppt == 3) && (group == 0 || group == 1), "union launch: unsupported accumulator width %d / group %d", ppt, group);
    if (ppt == 1) return group == 0 ? launch_union<1, 0>(d_table, totalBlocks, ldsBytes, st) : launch_union<1, 1>(d_table, totalBlocks, ldsBytes, st);
    return group == 0 ? launch_union<3, 0>(d_table, totalBlocks, ldsBytes, st) : launch_union<3, 1>(d_table, totalBlocks, ldsBytes, st);
}

size_t ne_mod2obs_lds_bytes(const ModelView &m, int d, int red) {
    const int fullDims = 6 + m.nJoints;
    size_t words = 24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8 + 2 * (size_t)m.nFrames + m.nSdfs + 4 * (size_t)m.nJoints;
    if (red) words += (size_t)NE_BLOCK * (fullDims | 1);
    return words * sizeof(float);
}

// launches the kernel only: partial systems land in slabs[nBlocks][S] (nBlocks chosen by the caller)
template <int RED, bool TRUNC>
static int launch_ne_mod2obs_kernel(int fullDims, int redDims, const float *d_pred, int width, int height,
                                    const ModelView &m, const float *d_T_gc, float trunc, const float *d_A,
                                    const int *d_map, float *slabs, int nBlocks, int *d_num, hipStream_t st,
                                    const int *d_nItems = nullptr) {
    const int d = RED ? redDims : fullDims;
    const int S = sys_size(d);
    const int nPix = width * height;
    const size_t ldsBytes = ne_mod2obs_lds_bytes(m, d, RED);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
    const float4 *pred = reinterpret_cast<const float4 *>(d_pred);
    const FrameBatch fb = cur_batch();
    const int tgcStride = (d_T_gc == m.T_mc) ? m.poseStride : fb.tab;      // the model's own (per-frame) T_mc, or an entry of the per-frame table
#define DARTK_LAUNCH_M2O(PPT_)                                                                                  \
    do {                                                                                                        \
        auto kfn = ne_mod2obs_kernel<PPT_, RED, TRUNC>;                                                         \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocks, fb.n), dim3(NE_BLOCK), ldsBytes, st, pred, nPix, m.modelId, m, d_T_gc, trunc, \
                           redDims, d_A, d_map, slabs, d_num, ReplayArgs{}, fb, tgcStride, d_nItems);           \
    } while (0)
    if (S <= NE_BLOCK) DARTK_LAUNCH_M2O(1);      // a small system: one output per thread, no idle accumulator slots
    else if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_M2O(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_M2O(6);
    else DARTK_LAUNCH_M2O(20);
#undef DARTK_LAUNCH_M2O
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

template <int RED, bool TRUNC>
static int launch_ne_mod2obs(int fullDims, int redDims, const float *d_pred, int width, int height, Model *mod,
                             const float *d_T_gc, float trunc, const float *d_A, const int *d_map, float *d_result,
                             int *d_num, bool zeroCounter, hipStream_t st, const int *d_nItems = nullptr) {
    const ModelView &m = mod->v;
    const int d = RED ? redDims : fullDims;
    DARTK_REQUIRE(fullDims == 6 + m.nJoints, "dims %d does not match the model's 6+%d", fullDims, m.nJoints);
    DARTK_REQUIRE(d >= 6 && d <= kMaxDims, "system dimension %d out of range [6,%d]", d, kMaxDims);
    DARTK_REQUIRE(width > 0 && height > 0, "bad prediction size %dx%d", width, height);
    DARTK_REQUIRE(mod->obsVoxels > 0, "model has no observation SDF");
    const int S = sys_size(d);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const int nPix = width * height;
    const int nTiles = (nPix + NE_BLOCK - 1) / NE_BLOCK;
    // a batch fills the chip with its frames; a frame's hit list is a few percent of its map
    const int cap = d_nItems ? 32 : cur_batch().n > 1 ? 128 : kMaxPartialBlocks;
    const int nBlocks = nTiles < cap ? nTiles : cap;
    rc = ensure_partials(ws, (size_t)cur_batch().n * nBlocks * S);
    if (rc) return rc;
    if (zeroCounter) DARTK_HIP(hipMemsetAsync(d_num, 0, sizeof(int), st));
    rc = launch_ne_mod2obs_kernel<RED, TRUNC>(fullDims, redDims, d_pred, width, height, m, d_T_gc, trunc, d_A, d_map,
                                              lane_partials(ws), nBlocks, d_num, st, d_nItems);
    if (rc) return rc;
    return launch_reduce_slabs(lane_partials(ws), nBlocks, S, d_result, 1, st);
}

// whole term (kernel + reduction into d_result) for the device-resident multi-model loop; red: 0 full, 1 linear, 2 map
int gn_ne_mod2obs_any(int red, int fullDims, int redDims, const float *d_pred, int width, int height, Model *mod,
                      const float *d_T_gc, const float *d_A, const int *d_map, float *d_result, int *d_num, hipStream_t st,
                      const int *d_nItems) {
    if (red == 1) return launch_ne_mod2obs<1, false>(fullDims, redDims, d_pred, width, height, mod, d_T_gc, 0.f, d_A, nullptr, d_result, d_num, false, st, d_nItems);
    if (red == 2) return launch_ne_mod2obs<2, false>(fullDims, redDims, d_pred, width, height, mod, d_T_gc, 0.f, nullptr, d_map, d_result, d_num, false, st, d_nItems);
    return launch_ne_mod2obs<0, false>(fullDims, fullDims, d_pred, width, height, mod, d_T_gc, 0.f, nullptr, nullptr, d_result, d_num, false, st, d_nItems);
}

// entry points for the device-resident Gauss-Newton loop (dartk_gn.hip)
int gn_launch_ne_mod2obs(const float *d_pred, int width, int height, const ModelView &m, const float *d_T_gc,
                         float *slabs, int nBlocks, int *d_num, hipStream_t st) {
    return launch_ne_mod2obs_kernel<0, false>(6 + m.nJoints, 6 + m.nJoints, d_pred, width, height, m, d_T_gc, 0.f, nullptr,
                                              nullptr, slabs, nBlocks, d_num, st);
}
int gn_launch_raycast(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                      float *d_prediction, hipStream_t st) {
    return launch_raycast(flx, fly, ppx, ppy, width, height, modelNum, m, d_prediction, 0.0f, st);
}
// the two halves apart, with the model's slice of the trace buffer named by the caller: the traces of several models are
// independent and may run side by side; the replays write one prediction map and keep the model order
int gn_raycast_reserve(size_t floats) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    return ensure_ray_buf(ws, floats);
}
int gn_launch_raycast_trace_at(float flx, float fly, float ppx, float ppy, int width, int height, const ModelView &m, size_t bufOffset,
                               hipStream_t st) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    DARTK_REQUIRE((cur_batch().n - 1) * cur_batch().trace + bufOffset + ray_trace_floats(width, height, m.nSdfs) <= ws->rayFloats, "trace buffer was not reserved");
    return launch_raycast_trace(flx, fly, ppx, ppy, width, height, m, 0.0f, ws, st, bufOffset);
}
int gn_launch_raycast_replay_cull_multi(float flx, float fly, float ppx, float ppy, int width, int height, int nModels,
                                        Model *const *mods, const size_t *bufOffsets, float *d_prediction, const float *d_obs, int ow,
                                        int oh, hipStream_t st, int *d_hitCount, const float *d_tables) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    DARTK_REQUIRE(nModels >= 1 && nModels <= DARTK_MAX_MODELS, "bad model count %d", nModels);
    DARTK_REQUIRE(ow >= 2 && oh >= 2, "observation map too small");
    ReplayMultiArgs a;
    a.n = nModels;
    for (int m = 0; m < nModels; ++m) { a.m[m] = mods[m]->v; a.off[m] = bufOffsets[m]; }
    const size_t nPix = (size_t)width * height;
    size_t ldsFloats = 0;
    for (int m = 0; m < nModels; ++m) ldsFloats += 28 * (size_t)mods[m]->v.nSdfs + 1;
    ProfScope ps2(DARTK_PROF_RAYCAST_REPLAY, st);
    const int nChunks = (int)((nPix + 255) / 256);
    if (cur_batch().n > 1 && ldsFloats * sizeof(float) <= 64 * 1024) {
        // a batch: a few thousand workgroups in all, each walking many chunks of its frame with the tables staged once
        static const char *env = getenv("DARTK_REPLAY_WGS");     // diagnostic override of the launch's workgroup budget
        int perFrame = (env ? atoi(env) : 2048) / cur_batch().n;
        perFrame = perFrame < 1 ? 1 : (perFrame > nChunks ? nChunks : perFrame);
        hipLaunchKernelGGL(raycast_replay_multi_rows_kernel, dim3(perFrame, cur_batch().n), dim3(256), ldsFloats * sizeof(float), st, flx, fly,
                           ppx, ppy, width, height, a, (const float *)ws->rayBuf, (float4 *)d_prediction, (const float4 *)d_obs, ow, oh, cur_batch(),
                           d_hitCount, (const float *)nullptr, 0);
    } else if (d_tables && cur_batch().n == 1 && ldsFloats * sizeof(float) <= 64 * 1024) {
        // a single frame whose tables the launch before has left in memory: a workgroup per 256-pixel chunk, the caller's map kept
        DARTK_REQUIRE(!d_hitCount, "the hit-list form of the replay is the batch's");
        hipLaunchKernelGGL(raycast_replay_multi_rows_kernel, dim3(nChunks, 1), dim3(256), ldsFloats * sizeof(float), st, flx, fly,
                           ppx, ppy, width, height, a, (const float *)ws->rayBuf, (float4 *)d_prediction, (const float4 *)d_obs, ow, oh, cur_batch(),
                           (int *)nullptr, d_tables, 1);
    } else {
        DARTK_REQUIRE(!d_hitCount, "the hit-list form of the replay needs the batch kernel (gn_replay_hit_lists)");
        hipLaunchKernelGGL(raycast_replay_multi_kernel, dim3((unsigned)nChunks, cur_batch().n), dim3(256), 0, st, flx, fly, ppx, ppy, width, height,
                           a, (const float *)ws->rayBuf, (float4 *)d_prediction, (const float4 *)d_obs, ow, oh, cur_batch());
    }
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
// whether gn_launch_raycast_replay_cull_multi would take the batch kernel for these models (the one that can emit hit lists)
bool gn_replay_hit_lists(int nModels, Model *const *mods) {
    size_t ldsFloats = 0;
    for (int m = 0; m < nModels; ++m) ldsFloats += 28 * (size_t)mods[m]->v.nSdfs + 1;
    return cur_batch().n > 1 && ldsFloats * sizeof(float) <= 64 * 1024 && getenv("DARTK_NO_HIT_LISTS") == nullptr;
}
int gn_launch_raycast_replay_at(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                                float *d_prediction, size_t bufOffset, hipStream_t st) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const size_t nPix = (size_t)width * height;
    ProfScope ps2(DARTK_PROF_RAYCAST_REPLAY, st);
    hipLaunchKernelGGL(raycast_replay_kernel, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, st, flx, fly, ppx, ppy, width, height,
                       modelNum, m, (const float2 *)(ws->rayBuf + bufOffset), (float4 *)d_prediction, 0.0f);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
// ray-cast + mod->obs normal equations of the same prediction: the trace kernel, then ONE kernel that replays each pixel's
// traces (writing the prediction for the caller) and goes on with the pixel -- one launch less on the critical path
int gn_launch_raycast_ne_mod2obs(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                                 float *d_prediction, const float *d_T_gc, float *slabs, int nBlocks, int *d_num, hipStream_t st,
                                 hipEvent_t beforeNe, bool traceDone, const float *d_tables, float *d_batchTables) {
    // beforeNe: an event the normal-equation kernel (the first reader of the obs SDF and of T_gc) waits for; the
    // ray-cast needs neither and starts at once
    if (m.nSdfs > NE_BLOCK) {
        DARTK_REQUIRE(!traceDone, "the shared trace launch needs at most %d SDFs", NE_BLOCK);
        int rc = launch_raycast(flx, fly, ppx, ppy, width, height, modelNum, m, d_prediction, 0.0f, st);
        if (rc) return rc;
        if (beforeNe) DARTK_HIP(hipStreamWaitEvent(st, beforeNe, 0));
        ProfScope ps(DARTK_PROF_NE_MOD2OBS, st);
        return gn_launch_ne_mod2obs(d_prediction, width, height, m, d_T_gc, slabs, nBlocks, d_num, st);
    }
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    if (!traceDone) {
        rc = launch_raycast_trace(flx, fly, ppx, ppy, width, height, m, 0.0f, ws, st);
        if (rc) return rc;
    }
    if (beforeNe) DARTK_HIP(hipStreamWaitEvent(st, beforeNe, 0));
    const int d = 6 + m.nJoints, S = sys_size(d), nPix = width * height;
    const size_t ldsBytes = ne_mod2obs_lds_bytes(m, d, 0) + sizeof(float) * (28 * (size_t)m.nSdfs + 1);
    DARTK_REQUIRE(ldsBytes <= 160 * 1024, "LDS requirement %zu exceeds 160 KiB", ldsBytes);
    // a batch (dartk_optimize_pose_batch): grid.y = frame, T_gc in the per-frame table, the replay tables of every frame's pose
    // written once here instead of being rebuilt by each of the frame's hundreds of workgroups
    const FrameBatch fb = cur_batch();
    DARTK_REQUIRE(fb.n == 1 || !d_tables, "replay tables are per pose: a batch passes d_batchTables");
    const int tabStride = 28 * m.nSdfs + 1;
    if (fb.n > 1 && d_batchTables) {
        hipLaunchKernelGGL(replay_tables_kernel, dim3(fb.n), dim3(NE_BLOCK), sizeof(float) * tabStride, st, m, flx, fly, ppx, ppy, width, height,
                           d_batchTables, tabStride);
        d_tables = d_batchTables;
    }
    // a batch whose frames have one slab per tile: four tiles per workgroup
    const int nTilesM2o = (nPix + NE_BLOCK - 1) / NE_BLOCK;
    static const char *envTpw = getenv("DARTK_M2O_BATCH_TILES");       // diagnostic override
    const int tpwB = envTpw && atoi(envTpw) > 0 ? atoi(envTpw) : 4;
    const bool perTile = fb.n > 1 && nBlocks == nTilesM2o && tpwB > 1;
    const int gridX = perTile ? (nTilesM2o + tpwB - 1) / tpwB : nBlocks;
    const ReplayArgs ra = {flx, fly, ppx, ppy, width, modelNum, (const float2 *)ws->rayBuf, (float4 *)d_prediction, d_tables,
                           fb.n > 1 ? tabStride : 0, perTile ? 1 : 0};
    const int tgcStride = fb.n > 1 ? fb.tab : 0;
    ProfScope ps(DARTK_PROF_NE_MOD2OBS, st);
#define DARTK_LAUNCH_RM2O(PPT_)                                                                                 \
    do {                                                                                                        \
        auto kfn = ne_mod2obs_kernel<PPT_, 0, false, true>;                                                     \
        if (ldsBytes > 64 * 1024)                                                                               \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL(kfn, dim3(gridX, fb.n), dim3(NE_BLOCK), ldsBytes, st, (const float4 *)d_prediction, nPix, m.modelId, m, d_T_gc, 0.f, \
                           d, nullptr, nullptr, slabs, d_num, ra, fb, tgcStride, (const int *)nullptr); \
    } while (0)
    if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_RM2O(3);
    else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_RM2O(6);
    else DARTK_LAUNCH_RM2O(20);
#undef DARTK_LAUNCH_RM2O
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
int gn_launch_cull(float *d_pred, int pw, int ph, const float *d_obs, int ow, int oh, hipStream_t st) {
    hipLaunchKernelGGL(cull_unobservable_kernel, dim3((pw * ph + 255) / 256), dim3(256), 0, st, (float4 *)d_pred, pw, ph,
                       (const float4 *)d_obs, ow, oh);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}
int gn_obs_sdf_build(const float *d_vert, int width, int height, Model *mod, float focalLength, hipStream_t st, float *d_T_gc_out) {
    const ModelView &m = mod->v;
    const int n = (int)mod->obsVoxels;
    const FrameBatch fb = cur_batch();              // a batch: the frames' observation SDFs lie obsStride (= n) floats apart
    DARTK_REQUIRE(fb.n == 1 || m.obsStride == n, "batched observation SDFs must be contiguous");
    // the splat goes to the scratch volume and the transform's three passes (scratch -> obsData -> scratch -> obsData) leave the
    // result in place: no copy back (the reference copies, optimizer.cpp:215)
    hipLaunchKernelGGL(splat_obs_sdf_kernel, dim3((n + 255) / 256, fb.n), dim3(256), 0, st, (const float4 *)d_vert, width, height, m, focalLength, fb,
                       mod->dtTmp, d_T_gc_out);
    DARTK_HIP(hipGetLastError());
    return launch_distance_transform_3d(mod->dtTmp, m.obsData, m.obs.dimx, m.obs.dimy, m.obs.dimz, 1, st, fb.n);
}

// gpu_errorModToObs, modToObs.cu:649-680 (prediction used in camera coordinates, as the reference does)
__global__ void __launch_bounds__(256)
error_mod2obs_kernel(const float4 *__restrict__ pred, int nPix, ModelView m, double *__restrict__ blockSums) {
    __shared__ double red[256];
    const SdfDev obs = m.obs;
    double s = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nPix; i += gridDim.x * blockDim.x) {
        const float4 pv = pred[i];
        if (pv.z == 0) continue;
        f3 p; p.x = pv.x; p.y = pv.y; p.z = pv.z;
        const f3 pg = grid_coords(obs, p);
        if (!in_bounds_grad_interp(obs, pg)) continue;
        const float err = grid_value(m.obsData, obs.dimx, obs.dimy, pg) * obs.res;
        s += (double)(float)(0.5 * (double)err * (double)err);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) blockSums[blockIdx.x] = red[0];
}

// The optional debug maps of normEqnsModToObs / Truncated / Reduced / ParamMap (modToObs.cu:47-49,65,74,82; truncated :145-147,
// 168,178,186): per prediction pixel the label, the residual and the obs-SDF gradient, with the reference's clearing rule (label
// and residual only cleared by model 0, the gradient map always). A kernel of its own: the maps are visualisation aids
// (optimization.h debugModToObs*), the normal-equation kernels stay free of their stores.
template <bool TRUNC>
__global__ void __launch_bounds__(256)
mod2obs_debug_maps_kernel(const float4 *__restrict__ pred, int nPix, int modelNum, ModelView m, const float *__restrict__ T_gc_in,
                          float truncationDist, int *__restrict__ dbgDA, float *__restrict__ dbgErr, float4 *__restrict__ dbgNorm) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= nPix) return;
    if (dbgDA && modelNum == 0) dbgDA[index] = -1;
    if (dbgErr && modelNum == 0) dbgErr[index] = __int_as_float(0x7fc00000);
    if (dbgNorm) dbgNorm[index] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pc = pred[index];
    if (pc.z == 0) return;
    float T_gc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T_gc[i] = T_gc_in[i];
    const SdfDev obs = m.obs;
    f3 p3; p3.x = pc.x; p3.y = pc.y; p3.z = pc.z;
    const f3 pm = se3_apply(T_gc, p3);
    const f3 pg = grid_coords(obs, pm);
    if (!in_bounds_grad_interp(obs, pg)) return;
    const float residual = grid_value(m.obsData, obs.dimx, obs.dimy, pg) * obs.res;
    if (TRUNC && (residual >= truncationDist || residual < 0)) return;
    if (dbgErr) dbgErr[index] = residual;
    const int label = (int)roundf(pc.w);
    if ((label >> 16) != modelNum) return;
    if (dbgDA) dbgDA[index] = label;
    if (dbgNorm) {
        const f3 g = grid_gradient(m.obsData, obs.dimx, obs.dimy, pg);
        dbgNorm[index] = make_float4(g.x, g.y, g.z, 1.0f);
    }
}

DARTK_COUNTER_ACCESS(mod2obs)

}  // namespace dartk

using namespace dartk;

extern "C" {

#ifdef DARTK_STAMPS
// out[1024][8]: the last mod->obs launch's per-workgroup stamps
int dartk_debug_m2o_phases(long long *out) {
    DARTK_HIP(hipDeviceSynchronize());
    DARTK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m2oStamps), sizeof(unsigned long long) * 1024 * 8));
    return DARTK_OK;
}
int dartk_debug_ray_steps(long long *out3) {
    int mx = 0; unsigned long long st = 0, n = 0;
    DARTK_HIP(hipDeviceSynchronize());
    DARTK_HIP(hipMemcpyFromSymbol(&mx, HIP_SYMBOL(g_rayMaxSteps), sizeof(mx)));
    DARTK_HIP(hipMemcpyFromSymbol(&st, HIP_SYMBOL(g_raySteps), sizeof(st)));
    DARTK_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_rayMarched), sizeof(n)));
    out3[0] = mx; out3[1] = (long long)st; out3[2] = (long long)n;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, zero6[6] = {0, 0, 0, 0, 0, 0};
    DARTK_HIP(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_rayPhase), sizeof(ph)));
    for (int i = 0; i < 6; ++i) out3[3 + i] = (long long)ph[i];      // the caller passes 9 slots
    DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rayPhase), zero6, sizeof(zero6)));
    const int z = 0; const unsigned long long zz = 0;
    DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rayMaxSteps), &z, sizeof(z)));
    DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_raySteps), &zz, sizeof(zz)));
    DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rayMarched), &zz, sizeof(zz)));
    return DARTK_OK;
}
#endif

int dartk_raycast_prediction(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                             dartk_model_t h, float *d_prediction, float levelSet, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_prediction, "dartk_raycast_prediction: null argument");
    DARTK_REQUIRE(width > 0 && height > 0 && flx != 0 && fly != 0, "bad camera %dx%d f=(%g,%g)", width, height, flx, fly);
    DARTK_REQUIRE(modelNum >= 0 && modelNum < 32768, "modelNum %d out of range", modelNum);
    return launch_raycast(flx, fly, ppx, ppy, width, height, modelNum, mod->v, d_prediction, levelSet, as_stream(stream));
}

int dartk_cull_unobservable(float *d_pred, int pw, int ph, const float *d_obs, int ow, int oh, void *stream) {
    DARTK_REQUIRE(d_pred && d_obs && pw > 0 && ph > 0 && ow > 1 && oh > 1, "dartk_cull_unobservable: bad argument");
    // the 2x2 neighbourhood read at (ox+1, oy+1) must stay inside the observation map
    DARTK_REQUIRE((long long)(pw - 1) * ow / pw + 1 < ow && (long long)(ph - 1) * oh / ph + 1 < oh,
                  "prediction %dx%d must be at most half the observation %dx%d resolution", pw, ph, ow, oh);
    hipLaunchKernelGGL(cull_unobservable_kernel, dim3((pw * ph + 255) / 256), dim3(256), 0, as_stream(stream),
                       (float4 *)d_pred, pw, ph, (const float4 *)d_obs, ow, oh);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_distance_transform_3d(float *d_in, float *d_out, int width, int height, int depth, int takeSqrt, void *stream) {
    DARTK_REQUIRE(d_in && d_out && d_in != d_out, "dartk_distance_transform_3d: bad buffers");
    return launch_distance_transform_3d(d_in, d_out, width, height, depth, takeSqrt, as_stream(stream));
}

int dartk_signed_distance_transform_3d(float *d_in, float *d_out, int width, int height, int depth, int takeSqrt, void *stream) {
    DARTK_REQUIRE(d_in && d_out && d_in != d_out, "dartk_signed_distance_transform_3d: bad buffers");
    DARTK_REQUIRE(width >= 1 && height >= 1 && depth >= 1, "bad dimensions");
    hipStream_t st = as_stream(stream);
    const int n = width * height * depth;
    float *seedDT = nullptr;
    DARTK_HIP(hipMalloc((void **)&seedDT, sizeof(float) * (size_t)n));   // cold path (model load), like the reference (:532)
    float *seed = d_out;
    hipLaunchKernelGGL(dt_seed_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_in, seed, width, height, depth);
    int rc = launch_distance_transform_3d(seed, seedDT, width, height, depth, takeSqrt, st);
    if (!rc) rc = launch_distance_transform_3d(d_in, d_out, width, height, depth, takeSqrt, st);
    if (!rc) hipLaunchKernelGGL(dt_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_out, seedDT, d_in, n);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(seedDT);
    if (rc) return rc;
    DARTK_HIP(e);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_obs_sdf_build(const float *d_vert, int width, int height, dartk_model_t h, float focalLength, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert, "dartk_obs_sdf_build: null argument");
    DARTK_REQUIRE(mod->obsVoxels > 0, "model has no observation SDF");
    DARTK_REQUIRE(width > 0 && height > 0, "bad map size");
    return gn_obs_sdf_build(d_vert, width, height, mod, focalLength, as_stream(stream));
}

static int upload_T(const float *h_T, float **d_T, hipStream_t st);

int dartk_splat_obs_sdf_zeros(const float *d_vert, int width, int height, const float *h_T_cm, dartk_model_t h, float focalLength,
                              void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && h_T_cm, "dartk_splat_obs_sdf_zeros: null argument");
    DARTK_REQUIRE(mod->obsVoxels > 0, "model has no observation SDF");
    DARTK_REQUIRE(width > 0 && height > 0, "bad map size");
    hipStream_t st = as_stream(stream);
    float *d_T = nullptr;
    int rc = upload_T(h_T_cm, &d_T, st);
    if (rc) return rc;
    ModelView m = mod->v;
    m.T_cm = d_T;               // the argument steers this launch, the model's own pose stays as it is (T_cm by reference there)
    const int n = (int)mod->obsVoxels;
    hipLaunchKernelGGL(splat_obs_sdf_kernel, dim3((n + 255) / 256, 1), dim3(256), 0, st, (const float4 *)d_vert, width, height, m, focalLength,
                       FrameBatch{1, 0, 0, 0, 0, 0, 0, 0}, mod->v.obsData, (float *)nullptr);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

static int upload_T(const float *h_T, float **d_T, hipStream_t st) {
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    // staged through the workspace scratch; synchronous (12 floats) so the host array may be a temporary
    DARTK_HIP(hipMemcpyAsync(ws->sysScratch, h_T, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    DARTK_HIP(hipStreamSynchronize(st));
    *d_T = ws->sysScratch;
    return DARTK_OK;
}

int dartk_ne_mod2obs(int dims, const float *d_pred, int width, int height, dartk_model_t h, const float *h_T_gc,
                     float *d_result, int32_t *d_num, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && h_T_gc && d_result && d_num, "dartk_ne_mod2obs: null argument");
    float *d_T = nullptr;
    int rc = upload_T(h_T_gc, &d_T, as_stream(stream));
    if (rc) return rc;
    return launch_ne_mod2obs<0, false>(dims, dims, d_pred, width, height, mod, d_T, 0.f, nullptr, nullptr, d_result, d_num, true, as_stream(stream));
}

int dartk_ne_mod2obs_truncated(int dims, const float *d_pred, int width, int height, dartk_model_t h, float trunc,
                               float *d_result, int32_t *d_num, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && d_result && d_num, "dartk_ne_mod2obs_truncated: null argument");
    return launch_ne_mod2obs<0, true>(dims, dims, d_pred, width, height, mod, mod->v.T_mc, trunc, nullptr, nullptr, d_result, d_num, true, as_stream(stream));
}

int dartk_ne_mod2obs_reduced(int dims, int redDims, const float *d_A, const float *d_pred, int width, int height,
                             dartk_model_t h, float *d_result, int32_t *d_num, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && d_result && d_num && d_A, "dartk_ne_mod2obs_reduced: null argument");
    DARTK_REQUIRE(redDims >= 6 && redDims <= dims, "redDims %d must be in [6,%d]", redDims, dims);
    return launch_ne_mod2obs<1, false>(dims, redDims, d_pred, width, height, mod, mod->v.T_mc, 0.f, d_A, nullptr, d_result, d_num, false, as_stream(stream));
}

int dartk_ne_mod2obs_parammap(int dims, int redDims, const int32_t *d_map, const float *d_pred, int width, int height,
                              dartk_model_t h, float *d_result, int32_t *d_num, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && d_result && d_num && d_map, "dartk_ne_mod2obs_parammap: null argument");
    DARTK_REQUIRE(redDims >= 6 && redDims <= dims, "redDims %d must be in [6,%d]", redDims, dims);
    return launch_ne_mod2obs<2, false>(dims, redDims, d_pred, width, height, mod, mod->v.T_mc, 0.f, nullptr, d_map, d_result, d_num, false, as_stream(stream));
}

int dartk_mod2obs_debug_maps(const float *d_pred, int width, int height, dartk_model_t h, const float *h_T_gc, int truncated,
                             float truncationDistance, int32_t *d_dbgDA, float *d_dbgErr, float *d_dbgNorm, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && width > 0 && height > 0, "dartk_mod2obs_debug_maps: bad argument");
    DARTK_REQUIRE(mod->obsVoxels > 0, "model has no observation SDF");
    if (!d_dbgDA && !d_dbgErr && !d_dbgNorm) return DARTK_OK;
    hipStream_t st = as_stream(stream);
    const float *d_T = mod->v.T_mc;          // the Truncated / Reduced / ParamMap wrappers use the model's own transform
    if (h_T_gc) {
        float *up = nullptr;
        int rc = upload_T(h_T_gc, &up, st);
        if (rc) return rc;
        d_T = up;
    }
    const int nPix = width * height;
    if (truncated)
        hipLaunchKernelGGL(mod2obs_debug_maps_kernel<true>, dim3((nPix + 255) / 256), dim3(256), 0, st, (const float4 *)d_pred, nPix,
                           mod->v.modelId, mod->v, d_T, truncationDistance, d_dbgDA, d_dbgErr, (float4 *)d_dbgNorm);
    else
        hipLaunchKernelGGL(mod2obs_debug_maps_kernel<false>, dim3((nPix + 255) / 256), dim3(256), 0, st, (const float4 *)d_pred, nPix,
                           mod->v.modelId, mod->v, d_T, 0.0f, d_dbgDA, d_dbgErr, (float4 *)d_dbgNorm);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_error_mod2obs(const float *d_pred, int width, int height, dartk_model_t h, float *h_error, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_pred && h_error, "dartk_error_mod2obs: null argument");
    DARTK_REQUIRE(mod->obsVoxels > 0, "model has no observation SDF");
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int nBlocks = 64;
    double *sums = reinterpret_cast<double *>(ws->partials);
    hipLaunchKernelGGL(error_mod2obs_kernel, dim3(nBlocks), dim3(256), 0, st, (const float4 *)d_pred, width * height, mod->v, sums);
    DARTK_HIP(hipGetLastError());
    double hs[64];
    DARTK_HIP(hipMemcpyAsync(hs, sums, sizeof(hs), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipStreamSynchronize(st));
    double s = 0;
    for (int i = 0; i < nBlocks; ++i) s += hs[i];
    *h_error = (float)s;
    return DARTK_OK;
}

}  // extern "C"
