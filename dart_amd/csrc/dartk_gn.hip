// dartk_gn.hip -- device-resident Gauss-Newton loops: the host side of Optimizer::optimizePose
// (optimizer.cpp:423-522) with no host round trip inside an iteration. Per iteration the reference does: host
// FK + 2 blocking uploads, a blocking count download, a memset + kernel + blocking system download per term,
// Eigen assembly/solve on the host. Here an iteration is four launches on one stream
//     ray-cast prediction | fused association + obs->mod normal equations | mod->obs normal equations
//     | reduce + assemble + LDL^T solve + pose update + forward kinematics
// and the pose never leaves HBM.
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_fk.h"
#include "dartk_raycast.h"

#include <cstdlib>

#include <vector>

namespace dartk {

static constexpr int GN_REDUCE_SEGS = 16;     // slab segments summed in parallel by gn_reduce_kernel, then added in a fixed order
static constexpr int GN_MAX_BLOCKS = 1024;    // partial systems per term at most (the reduction kernel sums them)

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg) -------
struct ProfEvent { int cls; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfEvent> g_prof;
int prof_push(int cls, hipStream_t st) {
    if (!g_prof_on) return -1;
    ProfEvent e; e.cls = cls;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return -1;
    (void)hipEventRecord(e.a, st);
    g_prof.push_back(e);
    return (int)g_prof.size() - 1;
}
bool prof_enabled() { return g_prof_on; }
void prof_pop(int idx, hipStream_t st) { (void)hipEventRecord(g_prof[idx].b, st); }

// -----------------------------------------------------------------------------------------------------
// Fused association + obs->mod normal equations (obsToMod.cu:12-114 + :246-337). The reference materialises a
// DataAssociatedPoint list between two kernels and reads the count back to the host; here the associated pixel
// goes straight into an LDS row (the SDF gradient computed for the normal test is reused for the Jacobian) and
// only the per-workgroup partial system and a count leave the CU.
// -----------------------------------------------------------------------------------------------------
// The observation is fixed for the whole optimizePose call, so the pixels that can ever associate (vertex w > 0)
// are compacted once per call into an index list (deterministic: pixel order is kept), and every iteration's fused
// kernel runs one full tile of real work per workgroup instead of striding over a mostly empty image.
static constexpr int VC_BLOCK = 256;
// zero2 (optional): two counters the iteration's kernels expect at zero (this is the first kernel of the call on its stream)
// (both kernels: grid.y = frame of a batch; the frames' maps, counts, lists and totals lie nPix / gridDim.x / nPix / 1 apart)
// masks: [frames][ceil(nPix / 64)] validity bits, read by valid_write_kernel (the map itself is read once per call)
__global__ void __launch_bounds__(VC_BLOCK)
valid_count_kernel(const float4 *__restrict__ vert, int nPix, int *__restrict__ blockCounts, int *__restrict__ zero2,
                   unsigned long long *__restrict__ masks) {
    __shared__ int wc[VC_BLOCK / 64];
    vert += (size_t)blockIdx.y * nPix;
    blockCounts += (size_t)blockIdx.y * gridDim.x;
    masks += (size_t)blockIdx.y * ((nPix + 63) / 64);
    const int index = blockIdx.x * VC_BLOCK + threadIdx.x;
    const bool v = index < nPix && vert[index].w > 0;
    const unsigned long long mask = __ballot(v);
    if ((threadIdx.x & 63) == 0) {
        wc[threadIdx.x >> 6] = __popcll(mask);
        if (index < nPix) masks[index >> 6] = mask;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blockCounts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
        if (zero2 && blockIdx.x == 0 && blockIdx.y == 0) { zero2[0] = 0; zero2[1] = 0; }
    }
}
// every workgroup sums the counts of the workgroups before it (a few KB, integer sums: order-free) instead of waiting for a scan
// kernel of its own -- one launch less at the head of every call; the last workgroup leaves the total in *nValid
__global__ void __launch_bounds__(VC_BLOCK)
valid_write_kernel(const unsigned long long *__restrict__ masks, int nPix, const int *__restrict__ blockCounts, int *__restrict__ validIdx,
                   int *__restrict__ nValid) {
    __shared__ int wc[VC_BLOCK / 64];
    __shared__ int ws[VC_BLOCK / 64];
    masks += (size_t)blockIdx.y * ((nPix + 63) / 64);
    blockCounts += (size_t)blockIdx.y * gridDim.x;
    validIdx += (size_t)blockIdx.y * nPix;
    nValid += blockIdx.y;
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += VC_BLOCK) before += blockCounts[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = before;
    const int index = blockIdx.x * VC_BLOCK + threadIdx.x;
    const bool v = index < nPix && ((masks[index >> 6] >> (index & 63)) & 1ull);
    int total;
    const int slot = compact_slot<VC_BLOCK>(v, wc, total);      // (synchronises the workgroup)
    const int off = ws[0] + ws[1] + ws[2] + ws[3];
    if (v) validIdx[off + slot] = index;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *nValid = off + total;
}

// d_work: valid_pixel_work_ints(nPix, frames) ints of scratch; d_validIdx: [frames][nPix]; d_nValid: [frames]
int gn_valid_pixels(const float *d_vert, int nPix, int nFrames, int *d_work, int *d_validIdx, int *d_nValid, hipStream_t st) {
    const int nb = (nPix + VC_BLOCK - 1) / VC_BLOCK;
    unsigned long long *masks = valid_pixel_masks(d_work, nPix, nFrames);
    hipLaunchKernelGGL(valid_count_kernel, dim3(nb, nFrames), dim3(VC_BLOCK), 0, st, (const float4 *)d_vert, nPix, d_work, (int *)nullptr, masks);
    hipLaunchKernelGGL(valid_write_kernel, dim3(nb, nFrames), dim3(VC_BLOCK), 0, st, (const unsigned long long *)masks, nPix, d_work, d_validIdx, d_nValid);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

// One tile of NE_BLOCK valid pixels per workgroup; workgroups past the end of the list exit (their slab is never read).
struct FusedObs2ModArgs {
    const float4 *vert, *norm;
    const int *validIdx, *nValidPtr;
    float distThr, normThr, planeOffset, pnx, pny, pnz, huberDelta;
    float *slabs;
    int *count;
};
template <int PPT>
__device__ __forceinline__ void fused_obs2mod_tile(int tile, const FusedObs2ModArgs &fa, const ModelView &m, float *lds) {
    const float4 *__restrict__ vert = fa.vert, *__restrict__ norm = fa.norm;
    const int *__restrict__ validIdx = fa.validIdx;
    const float distThr = fa.distThr, normThr = fa.normThr, planeOffset = fa.planeOffset, pnx = fa.pnx, pny = fa.pny, pnz = fa.pnz;
    const float huberDelta = fa.huberDelta;
    float *__restrict__ slabs = fa.slabs;
    int *__restrict__ count = fa.count;
    const int nValid = *fa.nValidPtr;
    if (tile * NE_BLOCK >= nValid) return;
    const int d = 6 + m.nJoints;
    const int ld = row_len(d, false);
    float *sT_fm = lds;
    float *sT_mf = sT_fm + 12 * m.nFrames;
    float *rows = sT_mf + 12 * m.nFrames;
    int *waveCnt = reinterpret_cast<int *>(rows + NE_BLOCK * ld);
    stage_model(m, sT_fm, sT_mf, nullptr);
    float T_mc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T_mc[i] = m.T_mc[i];

    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, false);
    const int k = tile * NE_BLOCK + threadIdx.x;
    bool active = false;
    AssocResult r;
    f3 gf; gf.x = gf.y = gf.z = 0.f;
    int frame = 0;
    __syncthreads();        // staged transforms visible
    if (k < nValid) {
        const int index = validIdx[k];
        const float4 xc = vert[index];
        r = associate_point(m, xc, distThr, planeOffset, pnx, pny, pnz);
        if (r.grid >= 0) {
            frame = m.sdfFrame[r.grid];
            const SdfDev &s = m.sdfs[r.grid];
            const f3 xf = se3_apply_w(sT_fm + 12 * frame, r.xm.x, r.xm.y, r.xm.z, r.w);
            const f3 xg = grid_coords(s, xf);
            gf = sdf_gradient<CNT_ASSOC_GRAD>(m.grids, s, xg);
            // normal test (obsToMod.cu:83-96), same arithmetic as normal_test()
            active = true;
            const float4 nObs = norm[index];
            if (nObs.w > 0.0f) {
                float T_fc[12], T_cf[12], T_f[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T_f[i] = sT_fm[12 * frame + i];
                se3_mul(T_f, T_mc, T_fc);
                se3_invert(T_fc, T_cf);
                const float invLen = 1.0f / sqrtf(gf.x * gf.x + gf.y * gf.y + gf.z * gf.z + 0.0f * 0.0f);
                const float gw = 0.0f * invLen;
                const f3 n = se3_apply_w(T_cf, gf.x * invLen, gf.y * invLen, gf.z * invLen, gw);
                const float dp = (-1 * n.x) * nObs.x + (-1 * n.y) * nObs.y + (-1 * n.z) * nObs.z + (-1 * gw) * 0.0f;
                if (dp < normThr) active = false;
            }
        }
    }
    int nActive;
    const int slot = compact_slot<NE_BLOCK>(active, waveCnt, nActive);
    if (active) {
        float *row = rows + slot * ld;
        const f3 gm = se3_rotate(sT_mf + 12 * frame, gf);
        error_jacobian<false>(row, r.xm, r.w, m.depMask[frame], gm, m.nJoints, m.jointType, m.jointAxis, sT_fm, sT_mf);
        const float err = r.err;
        float rr, E;
        if (fabsf(err) < huberDelta) { rr = -err; E = (float)(0.5 * (double)rr * (double)rr); }
        else {
            rr = (err < 0) ? huberDelta : -huberDelta;
            E = (float)((double)huberDelta * ((double)fabsf(err) - 0.5 * (double)huberDelta));
        }
        row[d] = rr; row[d + 1] = E; row[d + 2] = 1.0f;
    }
    __syncthreads();
    if (nActive) acc.accumulate(rows, ld, nActive);
    acc.store(slabs + (size_t)tile * sys_size(d), sys_size(d));
    if (threadIdx.x == 0 && nActive) atomicAdd(count, nActive);
}
// (grid.y = frame of a batch: maps fb.vert float4 apart, index lists fb.pts ints, counters fb.counts ints, valid-pixel totals 1 int,
// slabs gridDim.x partial systems apart; one frame: all strides 0)
template <int PPT>
__global__ void __launch_bounds__(NE_BLOCK)
fused_obs2mod_kernel(FusedObs2ModArgs fa, ModelView m_in, FrameBatch fb) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    const size_t f = blockIdx.y;
    select_frame(m, (int)f);
    fa.vert += f * fb.vert; fa.norm += f * fb.vert;
    fa.validIdx += f * fb.pts; fa.nValidPtr += f;
    fa.count += f * fb.counts;
    fa.slabs += f * gridDim.x * sys_size(6 + m.nJoints);
    // (one workgroup per tile in a batch too: a grid capped below the tile count, each workgroup walking several tiles, was measured
    // slower at every cap -- 220 us against 160 us per launch of 64 frames of 640x480, caps from 96 to 2400)
    fused_obs2mod_tile<PPT>(blockIdx.x, fa, m, lds);
}

// The ray-cast trace and the obs->mod term in ONE launch. Both need only the pose the solve just wrote and neither needs the
// other, so they used to run side by side on two streams -- and every fork and join of the two streams cost the iteration's
// critical chain ~6 us of event handling (profiles/r02_timeline.txt), a quarter of a whole kernel. As one grid (workgroups
// [0, nTraceWgs) walk the trace tasks, the rest take one tile of valid pixels each) they still run side by side, the whole
// iteration stays on ONE stream, and kernels that follow each other on one stream start back to back.
struct TraceArgs { float flx, fly, ppx, ppy; int width, height; float2 *trace; int nTraceWgs; float *tables; };
template <int PPT>
__global__ void __launch_bounds__(NE_BLOCK)
trace_and_obs2mod_kernel(TraceArgs ta, FusedObs2ModArgs fa, ModelView m) {
    extern __shared__ float lds[];
    if (blockIdx.x == gridDim.x - 1) {
        // one more workgroup leaves this pose's replay tables (per-SDF ray transforms, tile rectangles) in memory for the mod->obs
        // kernel that follows: its hundreds of workgroups then load 28 nSdfs + 1 words instead of each rebuilding them behind three
        // barriers -- that prologue was most of what that kernel's launch cost
        replay_tables_to_global(m, ta.flx, ta.fly, ta.ppx, ta.ppy, ta.width, ta.height, lds, ta.tables);
        return;
    }
    if ((int)blockIdx.x < ta.nTraceWgs) {       // the trace's stragglers are the launch's critical path: first in the grid
        int *sR = reinterpret_cast<int *>(lds);
        stage_ray_rects(m, 0.0f, ta.flx, ta.fly, ta.ppx, ta.ppy, ta.width, ta.height, sR);
        constexpr int W = NE_BLOCK / 64;
        raycast_trace_tasks(blockIdx.x * W + (threadIdx.x >> 6), ta.nTraceWgs * W, sR, ta.flx, ta.fly, ta.ppx, ta.ppy, ta.width, ta.height, m,
                            ta.trace, 0.0f);
        return;
    }
    fused_obs2mod_tile<PPT>(blockIdx.x - ta.nTraceWgs, fa, m, lds);
}

// -----------------------------------------------------------------------------------------------------
// reduce + assemble + solve + update (optimizer.cpp:462-518), one workgroup.
// -----------------------------------------------------------------------------------------------------
// One term of the normal equations: per-workgroup partial systems -> one packed system.
struct ReduceTerm {
    const float *slabs; int nBlocks; int S;
    const int *count; int minCount;      // system kept only if *count > minCount (obsToMod.cu:646); count may be null
    const int *nItems; int itemsPerBlock; // when set: only the first ceil(*nItems / itemsPerBlock) slabs were written
    float *out;
};
// grid.z = frame of a batch: a frame's slabs follow the previous frame's (nBlocks * S floats), its count lies countStride ints, its
// item total 1 int and its output outStride floats after the previous frame's
struct ReduceArgs { ReduceTerm t[4]; int countStride, outStride; };

// Fixed-order fp64 sum of the partial systems, spread over the chip: workgroup (x, term) owns 64 outputs of one
// term; its 4 waves each sum a quarter of the slabs and the quarters are combined in a fixed order. Bitwise
// reproducible (no atomics), ~1/4 MB read by ~15 workgroups instead of by one.
__global__ void __launch_bounds__(64 * GN_REDUCE_SEGS)
gn_reduce_kernel(ReduceArgs a) {
    __shared__ double part[GN_REDUCE_SEGS][64];
    ReduceTerm t = a.t[blockIdx.y];
    {
        const size_t f = blockIdx.z;
        if (t.slabs) t.slabs += f * (size_t)t.nBlocks * t.S;
        if (t.count) t.count += f * a.countStride;
        if (t.nItems) t.nItems += f;
        t.out += f * a.outStride;
    }
    const int o = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= t.S) return;
    const bool keep = t.slabs && (!t.count || *t.count > t.minCount);
    const int nBlocks = t.nItems ? min(t.nBlocks, (*t.nItems + t.itemsPerBlock - 1) / t.itemsPerBlock) : t.nBlocks;
    double s = 0.0;
    if (keep && o < t.S) {
        const int per = (nBlocks + GN_REDUCE_SEGS - 1) / GN_REDUCE_SEGS;
        const int b0 = q * per, b1 = min(nBlocks, b0 + per);
        const float *p = t.slabs + (size_t)b0 * t.S + o;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)k * t.S];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
            p += (size_t)8 * t.S;
        }
        for (; b < b1; ++b) { s += (double)*p; p += t.S; }
    }
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && o < t.S) {        // fixed order
        double tot = part[0][o & 63];
#pragma unroll
        for (int k = 1; k < GN_REDUCE_SEGS; ++k) tot += part[k][o & 63];
        t.out[o] = (float)tot;
    }
}

struct SolveArgs {
    const float *sysA; int *countA;     // obs->mod (reduced packed systems, see gn_reduce_kernel)
    const float *sysB; int *countB;     // mod->obs
    const float *sysC;                  // self-intersection (articulation-only system)
    float lambdaA, lambdaB, lambdaC;
    float regularization, lmLambda;
    float *iterInfo;        // 8 floats or null
    float *firstSystem;     // 3 packed systems or null
    int sysStride, countStride, infoStride;     // a batch (blockIdx.x = frame): floats / ints / floats between two frames; else 0
};

__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Pivoted LDL^T + solve by ONE wave on an LDS matrix (row stride ld). Same operation order as the oracle's
// restatement of Eigen's ldlt().solve (oracle/dart_oracle.c orc_solve_ldlt). x may alias b.
__device__ void ldlt_solve_wave(float *A, int ld, int n, const float *b, float *x, int *perm, float *y) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < n; i += 64) perm[i] = i;
    wave_sync();
    for (int k = 0; k < n; ++k) {
        // pivot = first index of the largest |diagonal| in [k, n)
        float best = -1.0f; int piv = n;
        for (int i = k + lane; i < n; i += 64) {
            const float v = fabsf(A[i * ld + i]);
            if (v > best) { best = v; piv = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int op = __shfl_xor(piv, o);
            if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
        }
        if (piv != k) {
            for (int j = lane; j < n; j += 64) { const float t = A[k * ld + j]; A[k * ld + j] = A[piv * ld + j]; A[piv * ld + j] = t; }
            wave_sync();
            for (int i = lane; i < n; i += 64) { const float t = A[i * ld + k]; A[i * ld + k] = A[i * ld + piv]; A[i * ld + piv] = t; }
            if (lane == 0) { const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
            wave_sync();
        }
        const float dk = A[k * ld + k];
        if (dk == 0.0f) continue;          // wave-uniform
        for (int i = k + 1 + lane; i < n; i += 64) { const float l = A[i * ld + k] / dk; A[i * ld + k] = l; y[i] = l; }
        wave_sync();
        // trailing update: lane j owns COLUMN j and walks the rows; L[.][k] is read from its copy in y (broadcast reads
        // that do not alias the column being written, so the loads pipeline). The oracle updates the lower triangle as
        // A[i][j] -= (L[i][k]*dk)*L[j][k] (i >= j) and mirrors it, so the upper entries take the mirrored product order.
        {
            const float *__restrict__ Lk = y;
            for (int j = k + 1 + lane; j < n; j += 64) {
                const float lj = Lk[j];
                float *__restrict__ col = A + j;
#pragma unroll 4
                for (int i = k + 1; i < n; ++i) {
                    const float li = Lk[i];
                    const float prod = (i >= j) ? (li * dk) * lj : (lj * dk) * li;
                    col[i * ld] = col[i * ld] - prod;
                }
            }
        }
        wave_sync();
    }
    for (int i = lane; i < n; i += 64) y[i] = b[perm[i]];
    wave_sync();
    for (int j = 0; j < n; ++j) {          // forward substitution, column oriented (same op order as row oriented)
        const float yj = y[j];
        for (int i = j + 1 + lane; i < n; i += 64) y[i] -= A[i * ld + j] * yj;
        wave_sync();
    }
    for (int i = lane; i < n; i += 64) {
        const float dd = A[i * ld + i];
        y[i] = (fabsf(dd) > 1.17549435e-38f) ? y[i] / dd : 0.0f;
    }
    wave_sync();
    for (int j = n - 1; j >= 0; --j) {     // backward substitution with L^T, column oriented
        const float yj = y[j];
        for (int i = lane; i < j; i += 64) y[i] -= A[j * ld + i] * yj;
        wave_sync();
    }
    for (int i = lane; i < n; i += 64) x[perm[i]] = y[i];
    wave_sync();
}

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// wave-wide maximum of a 32-bit key by DPP (gfx9 row operations), result uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast15 -> rows 1,3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// LDL^T with diagonal pivoting + solve for n <= 64 by ONE wave, lane j = unknown j. The symmetric matrix stays in LDS
// (row stride ld, odd) and is never permuted: step k picks the not-yet-eliminated index p with the largest |diagonal|
// (the pivot Eigen's ldlt() / the oracle pick, ties to the lowest index), scales row p into L and applies the rank-1
// update to the remaining columns, lane j walking column j. Per step: one DPP arg-max, one division per lane, and
// 2 LDS reads + 1 write per remaining row. The substitutions run in pivot order on a per-lane register. x[j] = lane j.
__device__ void ldlt_solve_wave64(float *A, int ld, int n, const float *b, float *x, int *order, float *Lk) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < n;
    float dj = mine ? A[lane * ld + lane] : 0.0f;     // running diagonal of this lane's index
    float dpiv = 0.0f;                                // pivot value once this index has been eliminated
    unsigned long long todo = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    for (int k = 0; k < n; ++k) {
        const bool open = mine && ((todo >> lane) & 1ull);
        const unsigned key = open ? (__float_as_uint(fabsf(dj)) + 1u) : 0u;
        const unsigned best = wave_max_u32(key);
        const int p = __builtin_ctzll(__ballot(key == best));
        const float dk = readlane_f(dj, p);
        if (lane == 0) order[k] = p;
        if (lane == p) dpiv = dk;
        todo &= ~(1ull << p);
        if (dk == 0.0f) continue;          // wave-uniform; the oracle / Eigen leave the column untouched
        const bool rest = mine && ((todo >> lane) & 1ull);
        float lj = 0.0f;
        if (rest) {
            lj = A[p * ld + lane] / dk;
            A[p * ld + lane] = lj;         // row p is dead from here on: it keeps L[.][p]
            Lk[lane] = lj;
            dj = dj - (lj * dk) * lj;
        }
        wave_sync();
        if (rest) {
            float *__restrict__ col = A + lane;
            const float *__restrict__ L = Lk;
            for (unsigned long long msk = todo; msk; msk &= msk - 1ull) {
                const int i = __builtin_ctzll(msk);
                col[i * ld] = col[i * ld] - (L[i] * dk) * lj;
            }
        }
        wave_sync();
    }
    // forward substitution in pivot order: y_k = b[p_k] - sum_{m<k} L[p_k][p_m] y_m
    float yv = mine ? b[lane] : 0.0f;
    todo = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    for (int k = 0; k < n; ++k) {
        const int p = __builtin_amdgcn_readfirstlane(order[k]);
        const float yp = readlane_f(yv, p);
        todo &= ~(1ull << p);
        if (mine && ((todo >> lane) & 1ull)) yv -= A[p * ld + lane] * yp;
    }
    // D^-1 (Eigen's pseudo-inverse: tiny pivots give 0)
    yv = (fabsf(dpiv) > 1.17549435e-38f) ? yv / dpiv : 0.0f;
    // backward substitution with L^T: x_{p_k} = y_k - sum_{m>k} L[p_m][p_k] x_{p_m}
    unsigned long long doneMask = 0ull;
    for (int k = n - 1; k >= 0; --k) {
        const int p = __builtin_amdgcn_readfirstlane(order[k]);
        const float xp = readlane_f(yv, p);
        doneMask |= (1ull << p);
        if (mine && !((doneMask >> lane) & 1ull)) yv -= A[lane * ld + p] * xp;
    }
    if (mine) x[lane] = yv;
    wave_sync();
}

// LDL^T with diagonal pivoting + solve for n <= NMAX = 32 by ONE wave with the matrix in REGISTERS: lane j owns column j
// (c[i] = A[i][j]) and unknown j. Nothing is permuted: step k picks the not-yet-eliminated index p with the largest
// |diagonal| (the pivot Eigen's ldlt() / the oracle pick, ties to the lowest index), every open lane turns its entry of row p
// into L[j][p] (kept in c[p]) and subtracts (L[i][p]*d_p)*L[j][p] from its open rows, L[i][p]*d_p coming from lane i by
// v_readlane. A single wave issues one instruction every four cycles whatever it is, so the step is written to be short
// (no LDS round trips, ~3 instructions per open row) rather than wide. c[p] with the wave-uniform p is register-relative
// addressing. The forward substitution runs on the same registers; the backward one needs L transposed (lane j: L[.][j]
// of the rows eliminated after j), which goes once through LDS (T, row stride ldT).
template <int NMAX>
__device__ __forceinline__ float ldlt_solve_regs(float (&c)[NMAX], float dj, int n, float bj, float *T, int ldT, unsigned long long *stamps) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < n;
    float dpiv = 0.0f;                                // pivot value once this index has been eliminated
    int orderV = 0;                                   // lane k: index eliminated at step k
    const unsigned long long all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    unsigned long long todo = all;
    // the forward substitution rides along: y_p is final when p is eliminated, and the open lanes subtract L[j][p]*y_p
    // right away (the same terms in the same order as a separate pass in pivot order)
    float yv = mine ? bj : 0.0f;
    for (int k = 0; k < n; ++k) {
        const bool open = mine && ((todo >> lane) & 1ull);
        const unsigned key = open ? (__float_as_uint(fabsf(dj)) + 1u) : 0u;
        const unsigned best = wave_max_u32(key);
        const int p = __builtin_ctzll(__ballot(key == best));
        const float dk = readlane_f(dj, p);
        const float yp = readlane_f(yv, p);
        if (lane == k) orderV = p;
        if (lane == p) dpiv = dk;
        todo &= ~(1ull << p);
        if (dk == 0.0f) {                  // wave-uniform; the oracle / Eigen leave the column untouched
            if (mine && ((todo >> lane) & 1ull)) yv = yv - c[p] * yp;
            continue;
        }
        // branch-free rank-1 step: lanes that are not open any more carry l = 0, so their multiplier t = l*d_p is 0 and
        // the rows they stand for (and row p, which keeps L[.][p]) are left as they are (x - 0*l == x for finite l; a
        // non-finite factor is garbage in the branching formulation as well)
        const bool rest = mine && ((todo >> lane) & 1ull);
        const float cp = c[p];
        const float lj = rest ? cp / dk : 0.0f;
        c[p] = rest ? lj : cp;
        dj = rest ? dj - (lj * dk) * lj : dj;
        yv = rest ? yv - lj * yp : yv;
        const float t = lj * dk;
#pragma unroll
        for (int i0 = 0; i0 < NMAX; i0 += 8) {
            if (i0 < n) {
#pragma unroll
                for (int i = i0; i < i0 + 8; ++i) c[i] = c[i] - readlane_f(t, i) * lj;
            }
        }
    }
#ifdef DARTK_STAMPS
    stamps[1] = __builtin_amdgcn_s_memrealtime();
#endif
    // D^-1 (Eigen's pseudo-inverse: tiny pivots give 0)
    yv = (fabsf(dpiv) > 1.17549435e-38f) ? yv / dpiv : 0.0f;
    // transpose L through LDS: T[i][j] = c[i] of lane j; lane j then holds r[p] = T[j][p] = L[p][j] for p eliminated after j
    if (mine) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if (i < n) T[i * ldT + lane] = c[i];
    }
    wave_sync();
    float r[NMAX];
#pragma unroll
    for (int q = 0; q < NMAX; ++q) r[q] = (mine && q < n) ? T[lane * ldT + q] : 0.0f;
    wave_sync();
    // backward substitution with L^T: x_{p_k} = y_k - sum_{m>k} L[p_m][p_k] x_{p_m}
    unsigned long long doneMask = 0ull;
    for (int k = n - 1; k >= 0; --k) {
        const int p = __builtin_amdgcn_readlane(orderV, k);
        const float xp = readlane_f(yv, p);
        doneMask |= (1ull << p);
        if (mine && !((doneMask >> lane) & 1ull)) yv = yv - r[p] * xp;
    }
    return yv;
}

// entry (i, j) of the damped normal matrix: unpack (optimizer.cpp:50-70) in the reference's term order, symmetrise,
// regularise, LM-scale
__device__ __forceinline__ float assemble_entry(const SolveArgs &a, int d, int i, int j) {
    const int ad = d - 6;
    const int lo = i < j ? i : j, hi = i < j ? j : i;          // upper (lo,hi) <- packed[hi(hi+1)/2 + lo]
    float v = 0.0f;
    if (a.sysA) v += a.lambdaA * a.sysA[d + ((hi * (hi + 1)) >> 1) + lo];
    if (a.sysB) v += a.lambdaB * a.sysB[d + ((hi * (hi + 1)) >> 1) + lo];
    if (a.sysC && lo >= 6) {
        const int p = lo - 6, q = hi - 6;
        float t = 0.0f;
        t += a.lambdaC * a.sysC[ad + ((q * (q + 1)) >> 1) + p];
        v += t;
    }
    if (i == j) { v += a.regularization * 1.0f; v += a.lmLambda * v; }
    return v;
}
__device__ __forceinline__ float assemble_rhs(const SolveArgs &a, int j) {
    float v = 0.0f;
    if (a.sysA) v += a.lambdaA * a.sysA[j];
    if (a.sysB) v += a.lambdaB * a.sysB[j];
    if (a.sysC && j >= 6) { float t = 0.0f; t += a.lambdaC * a.sysC[j - 6]; v += t; }
    return v;
}

// d <= NMAX: the packed systems are combined into one packed system in LDS (coalesced reads; per entry the same sums in
// the same order as assemble_entry), lane j then takes column j into registers and solves; returns x[j].
// W: LDS scratch of at least max(sys_size(d), d * ldT) floats.
template <int NMAX>
__device__ __forceinline__ float assemble_solve_regs(const SolveArgs &a, int d, float *W, int ldT, unsigned long long *stamps) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < d;
    const int j = mine ? lane : 0;
    const int S = sys_size(d), ad = d - 6, Sa = sys_size(ad);
    constexpr int NC = (sys_size(NMAX - 6) + 63) / 64;
    float tc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {      // the third system's loads go out with the first two
        const int o = lane + 64 * q;
        tc[q] = (a.sysC && o < Sa - 1) ? a.sysC[o] : 0.0f;
    }
    {   // all loads of both systems first, then the sums: one memory round trip instead of one per 64 entries
        constexpr int NI = (sys_size(NMAX) + 63) / 64;
        float va[NI], vb[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int o = lane + 64 * q;
            va[q] = (a.sysA && o < S - 1) ? a.sysA[o] : 0.0f;
            vb[q] = (a.sysB && o < S - 1) ? a.sysB[o] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int o = lane + 64 * q;
            if (o < S - 1) {
                float v = 0.0f;
                if (a.sysA) v += a.lambdaA * va[q];
                if (a.sysB) v += a.lambdaB * vb[q];
                W[o] = v;
            }
        }
    }
    wave_sync();
    if (a.sysC) {
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int o = lane + 64 * q;
            if (o >= Sa - 1) continue;
            float t = 0.0f;
            t += a.lambdaC * tc[q];
            int target;
            if (o < ad) target = 6 + o;
            else {
                const int u = o - ad;
                int q = (int)((sqrtf(8.0f * (float)u + 1.0f) - 1.0f) * 0.5f);
                while (((q * (q + 1)) >> 1) > u) --q;
                while ((((q + 1) * (q + 2)) >> 1) <= u) ++q;
                const int p = u - ((q * (q + 1)) >> 1);
                target = d + (((q + 6) * (q + 7)) >> 1) + p + 6;
            }
            W[target] += t;
        }
        wave_sync();
    }
    float c[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;          // upper (lo,hi) <- packed[hi(hi+1)/2 + lo]
        c[i] = (i < d) ? W[d + ((hi * (hi + 1)) >> 1) + lo] : 0.0f;
    }
    float dj = c[0];
#pragma unroll
    for (int i = 1; i < NMAX; ++i) dj = (i == j) ? c[i] : dj;
    dj += a.regularization * 1.0f;
    dj += a.lmLambda * dj;
    const float bj = W[j];
    wave_sync();
#ifdef DARTK_STAMPS
    stamps[0] = __builtin_amdgcn_s_memrealtime();
#endif
    return ldlt_solve_regs<NMAX>(c, dj, d, bj, W, ldT, stamps);
}

// assemble + solve + update + forward kinematics (optimizer.cpp:482-518), ONE wave.
__global__ void __launch_bounds__(64)
gn_solve_single_kernel(ModelView m_in, SolveArgs a_in) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    SolveArgs a = a_in;
    {   // frame of a batch (one wave per frame); a single call: frame 0, strides 0
        const size_t f = blockIdx.x;
        select_frame(m, (int)f);
        if (a.sysA) a.sysA += f * a.sysStride;
        if (a.sysB) a.sysB += f * a.sysStride;
        if (a.sysC) a.sysC += f * a.sysStride;
        a.countA += f * a.countStride; a.countB += f * a.countStride;
        if (a.iterInfo) a.iterInfo += f * a.infoStride;
    }
    const int d = 6 + m.nJoints;
    const int ad = d - 6;
    const int S = sys_size(d), Sa = sys_size(ad);
    const int ld = (d + 1) | 1;             // odd row stride: row and column walks are both bank-conflict free
    float *JTJ = lds;                       // [d][ld]
    float *eJ = JTJ + d * ld;
    float *x = eJ + d;
    float *y = x + d;
    int *perm = reinterpret_cast<int *>(y + d);
    float *sT = reinterpret_cast<float *>(perm + d);      // forward-kinematics scratch, 12*nFrames
    const int tid = threadIdx.x;
    const float *__restrict__ sysA = a.sysA, *__restrict__ sysB = a.sysB, *__restrict__ sysC = a.sysC;
#ifdef DARTK_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
    // everything the update and the forward kinematics read that does not depend on the solution is requested now
    const FkTables fkT = fk_prefetch(m);
    const float jointOld = tid < m.nJoints ? m.joints[tid] : 0.0f;
    float Told[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Told[i] = m.T_mc[i];

    if (a.firstSystem) {
        for (int o = tid; o < S; o += 64) { a.firstSystem[o] = sysA ? sysA[o] : 0.0f; a.firstSystem[S + o] = sysB ? sysB[o] : 0.0f; }
        for (int o = tid; o < Sa; o += 64) a.firstSystem[2 * S + o] = sysC ? sysC[o] : 0.0f;
    }
    if (tid == 0) {
        if (a.iterInfo) {
            a.iterInfo[0] = sysA ? a.lambdaA * sysA[S - 1] : 0.0f;
            a.iterInfo[1] = sysB ? a.lambdaB * sysB[S - 1] : 0.0f;
            a.iterInfo[2] = sysC ? a.lambdaC * sysC[Sa - 1] : 0.0f;
            a.iterInfo[3] = sysA ? (float)*a.countA : 0.0f;
            a.iterInfo[4] = sysB ? (float)*a.countB : 0.0f;
            a.iterInfo[5] = a.iterInfo[6] = a.iterInfo[7] = 0.0f;
        }
        *a.countA = 0; *a.countB = 0;       // ready for the next iteration's kernels (stream order)
    }
#ifdef DARTK_STAMPS
    unsigned long long st1 = st0, stF = 0;
#endif
    if (d <= 32) {
        unsigned long long stamps[2] = {0, 0};
        const float xj = assemble_solve_regs<32>(a, d, JTJ, ld, stamps);
#ifdef DARTK_STAMPS
        st1 = stamps[0]; stF = stamps[1];
#endif
        if (tid < d) x[tid] = xj;
        wave_sync();
    } else {
        // unpack (optimizer.cpp:50-70) in the reference's term order, symmetrise, regularise, LM-scale
        for (int idx = tid; idx < d * d; idx += 64) {
            const int i = idx / d, j = idx % d;
            JTJ[i * ld + j] = assemble_entry(a, d, i, j);
        }
        for (int j = tid; j < d; j += 64) eJ[j] = assemble_rhs(a, j);
        wave_sync();
#ifdef DARTK_STAMPS
        st1 = __builtin_amdgcn_s_memrealtime();
#endif
        if (d <= 64) ldlt_solve_wave64(JTJ, ld, d, eJ, x, perm, y);
        else ldlt_solve_wave(JTJ, ld, d, eJ, x, perm, y);
    }
#ifdef DARTK_STAMPS
    const unsigned long long st2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid == 0) {
        float t6[6], dT[12], Tn[12], Tinv[12];
        for (int i = 0; i < 6; ++i) t6[i] = -x[i];
        se3_exp(t6, dT);
        se3_mul(dT, Told, Tn);
        se3_invert(Tn, Tinv);
        for (int i = 0; i < 12; ++i) { m.T_mc[i] = Tn[i]; m.T_cm[i] = Tinv[i]; }
    }
    float jointNew = 0.0f;
    if (tid < m.nJoints) {
        const float v = jointOld + (-x[6 + tid]);
        jointNew = fminf(fmaxf(fkT.lo, v), fkT.hi);
        m.joints[tid] = jointNew;
    }
    // forward kinematics for the next iteration (MirroredModel::setPose); lane j carries the joint it just wrote
    fk_wave(m, 0, sT, fkT, jointNew);
#ifdef DARTK_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && a.iterInfo) {   // diagnostic build only: 10 ns ticks of assemble / LDL^T / update+FK in the spare slots
        a.iterInfo[5] = (float)(st1 - st0); a.iterInfo[6] = (float)(st2 - st1); a.iterInfo[7] = stF ? (float)(stF - st1) : (float)(st3 - st2);
    }
#endif
}

static size_t solve_lds_bytes(int d, int nFrames) {
    return sizeof(float) * (size_t)(d * (d + 2) + 4 * d + 12 * nFrames + 8);
}

// dartk_optimize_pose_batch: poses in (mode 0: T_mc, its inverse, the joints, forward kinematics as MirroredModel::setPose does) and
// out (mode 1) of the frames' pose records; one wave per frame
__global__ void __launch_bounds__(64)
gn_batch_pose_kernel(ModelView m_in, float *__restrict__ T_mcs, float *__restrict__ joints, int mode) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    const int f = blockIdx.x, lane = threadIdx.x;
    select_frame(m, f);
    float *T = T_mcs + (size_t)f * 12;
    float *j = joints ? joints + (size_t)f * m.nJoints : nullptr;
    if (mode == 1) {
        if (lane < 12) T[lane] = m.T_mc[lane];
        if (lane < m.nJoints) j[lane] = m.joints[lane];
        return;
    }
    if (lane == 0) {
        float A[12], inv[12];
        for (int i = 0; i < 12; ++i) A[i] = T[i];
        se3_invert(A, inv);
        for (int i = 0; i < 12; ++i) { m.T_mc[i] = A[i]; m.T_cm[i] = inv[i]; }
    }
    const float value = lane < m.nJoints ? j[lane] : 0.0f;
    if (lane < m.nJoints) m.joints[lane] = value;
    const FkTables t = fk_prefetch(m);
    fk_wave(m, 0, lds, t, value);
}

// per-device scratch of the batched loop (grown on demand, kept until the process ends)
struct BatchScratch { void *p; size_t bytes; };
static BatchScratch g_batchScratch[16];
static int batch_scratch(size_t bytes, void **out) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    if (g_batchScratch[dev].bytes < bytes) {
        if (g_batchScratch[dev].p) { DARTK_HIP(hipDeviceSynchronize()); DARTK_HIP(hipFree(g_batchScratch[dev].p)); }
        g_batchScratch[dev].p = nullptr; g_batchScratch[dev].bytes = 0;
        DARTK_HIP(hipMalloc(&g_batchScratch[dev].p, bytes));
        g_batchScratch[dev].bytes = bytes;
    }
    *out = g_batchScratch[dev].p;
    return DARTK_OK;
}

// One chunk of dartk_optimize_pose_batch: nf frames side by side through the kernels of dartk_optimize_pose (grid.y / grid.z =
// frame; reduction and solve: per frame what the single call launches), on one stream.
static int optimize_pose_chunk(Model *mod, int nf, const float *d_vert, const float *d_norm, int width, int height, const dartk_opt_options *o,
                               const float *d_cloud, int nCloud, const int32_t *d_potential, float *d_T_mcs, float *d_joints,
                               float *d_iterInfo, hipStream_t st) {
    const bool useA = o->lambdaObsToMod > 0, useB = o->lambdaModToObs > 0;
    const bool useC = o->lambdaIntersection > 0 && nCloud > 0;
    const ModelView &m0 = mod->v;
    const int d = 6 + m0.nJoints, S = sys_size(d), Sa = sys_size(d - 6);
    const int nPix = width * height, nPred = useB ? o->predWidth * o->predHeight : 0;
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    // the partial systems of a frame: the same number of slabs per term as the single call (the sums keep their order)
    const int nBlocksA = (nPix + NE_BLOCK - 1) / NE_BLOCK;
    const int tilesB = (nPred + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocksB = tilesB < GN_MAX_BLOCKS ? tilesB : GN_MAX_BLOCKS;
    const int tilesC = (nCloud + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocksC = tilesC < GN_MAX_BLOCKS ? tilesC : GN_MAX_BLOCKS;
    const size_t slabA = useA ? (size_t)nBlocksA * S : 0, slabB = useB ? (size_t)nBlocksB * S : 0, slabC = useC ? (size_t)nBlocksC * Sa : 0;
    rc = ensure_partials(ws, (size_t)nf * (slabA + slabB + slabC));
    if (rc) return rc;
    float *slabsA = ws->partials, *slabsB = slabsA + nf * slabA, *slabsC = slabsB + nf * slabB;
    const int nVcBlocks = (nPix + VC_BLOCK - 1) / VC_BLOCK;
    rc = ensure_valid_idx(ws, (size_t)nf * ((size_t)nPix + 1) + valid_pixel_work_ints(nPix, nf));
    if (rc) return rc;
    int *validIdx = ws->validIdx, *vcWork = validIdx + (size_t)nf * nPix, *d_nValid = vcWork + valid_pixel_work_ints(nPix, nf);
    (void)nVcBlocks;

    // one allocation: [pose records | observation SDFs | distance-transform scratch | prediction maps | T_gc | systems | counters]
    const size_t rec = 24 + 24 * (size_t)m0.nFrames + (size_t)(m0.nJoints > 0 ? m0.nJoints : 1);
    const int sysStride = 2 * S + Sa;
    size_t total = 0;
    auto take = [&](size_t bytes) { size_t at = (total + 255) / 256 * 256; total = at + bytes; return at; };
    const size_t o_pose = take(sizeof(float) * rec * nf);
    const size_t o_obs = take(sizeof(float) * mod->obsVoxels * nf * (useB ? 1 : 0));
    const size_t o_tmp = take(sizeof(float) * mod->obsVoxels * nf * (useB ? 1 : 0));
    const size_t o_pred = take(256);        // (the batch returns no prediction maps: the replay keeps each pixel in registers)
    const size_t o_tgc = take(sizeof(float) * 12 * nf);
    const size_t o_sys = take(sizeof(float) * (size_t)sysStride * nf);
    const size_t o_cnt = take(sizeof(int) * 2 * (size_t)nf);
    const size_t o_rtab = take(sizeof(float) * (28 * (size_t)m0.nSdfs + 1) * nf * (useB ? 1 : 0));
    void *base = nullptr;
    rc = batch_scratch(total, &base);
    if (rc) return rc;
    auto at = [&](size_t off) { return reinterpret_cast<float *>((char *)base + off); };
    Model view = *mod;          // the handle's static tables with per-frame pose / observation-SDF arrays
    ModelView &m = view.v;
    float *p = at(o_pose);
    m.T_mc = p; m.T_cm = p + 12; m.T_mf = p + 24; m.T_fm = p + 24 + 12 * m.nFrames; m.joints = p + 24 + 24 * m.nFrames;
    m.poseStride = (int)rec;
    if (useB) { m.obsData = at(o_obs); m.obsStride = (int)mod->obsVoxels; view.dtTmp = at(o_tmp); }
    float *d_pred = nullptr, *d_T_gc = at(o_tgc);
    (void)o_pred;
    float *sysA = at(o_sys), *sysB = sysA + S, *sysC = sysB + S;
    int *cntA = reinterpret_cast<int *>((char *)base + o_cnt), *cntB = cntA + 1;
    const size_t traceFloats = useB ? ray_trace_floats(o->predWidth, o->predHeight, m.nSdfs) : 0;
    const FrameBatch fb = {nf, 2, sysStride, 12, (size_t)nPix, (size_t)nPred, traceFloats, (size_t)nPix};
    BatchScope scope(fb);

    const size_t ldsS = solve_lds_bytes(d, m.nFrames);
    const size_t ldsA = sizeof(float) * (24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8);
    DARTK_REQUIRE(ldsA <= 160 * 1024 && ldsS <= 160 * 1024, "LDS requirement exceeds 160 KiB");
    if (ldsS > 64 * 1024)
        DARTK_HIP(hipFuncSetAttribute((const void *)gn_solve_single_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsS));
    hipLaunchKernelGGL(gn_batch_pose_kernel, dim3(nf), dim3(64), sizeof(float) * 12 * (size_t)(m.nFrames + 1), st, m, d_T_mcs, d_joints, 0);
    DARTK_HIP(hipGetLastError());
    DARTK_HIP(hipMemsetAsync(cntA, 0, sizeof(int) * 2 * (size_t)nf, st));
    if (useB) {     // generateObsSdf with every frame's start pose; T_obsSdf_camera frozen (optimizer.cpp:445-449)
        ProfScope ps(DARTK_PROF_OBS_SDF, st);
        rc = gn_obs_sdf_build(d_vert, width, height, &view, o->focalLength, st, nullptr);
        if (rc) return rc;
        DARTK_HIP(hipMemcpy2DAsync(d_T_gc, sizeof(float) * 12, m.T_mc, sizeof(float) * rec, sizeof(float) * 12, nf, hipMemcpyDeviceToDevice, st));
        rc = gn_raycast_reserve(traceFloats * nf);
        if (rc) return rc;
    }
    if (useA && o->numIterations > 0) {
        ProfScope ps(DARTK_PROF_OTHER, st);
        rc = gn_valid_pixels(d_vert, nPix, nf, vcWork, validIdx, d_nValid, st);
        if (rc) return rc;
    }
    const FusedObs2ModArgs fa = {(const float4 *)d_vert, (const float4 *)d_norm, validIdx, d_nValid, o->distThreshold, o->normThreshold,
                                 o->planeOffset, o->planeNormal[0], o->planeNormal[1], o->planeNormal[2], o->huberDelta, slabsA, cntA};
    for (int it = 0; it < o->numIterations; ++it) {
        if (useB) {
            rc = gn_launch_raycast_ne_mod2obs(o->predFl[0], o->predFl[1], (float)(o->predWidth / 2), (float)(o->predHeight / 2), o->predWidth,
                                              o->predHeight, 0, m, d_pred, d_T_gc, slabsB, nBlocksB, cntB, st, nullptr, false, nullptr, at(o_rtab));
            if (rc) return rc;
        }
        if (useA) {
            ProfScope ps(DARTK_PROF_FUSED_OBS2MOD, st);
#define DARTK_LAUNCH_FUSED_B(PPT_)                                                                              \
    do {                                                                                                        \
        auto kfn = fused_obs2mod_kernel<PPT_>;                                                                  \
        if (ldsA > 64 * 1024)                                                                                   \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA)); \
        hipLaunchKernelGGL(kfn, dim3(nBlocksA, nf), dim3(NE_BLOCK), ldsA, st, fa, m, fb);                       \
    } while (0)
            if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_FUSED_B(3);
            else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_FUSED_B(6);
            else DARTK_LAUNCH_FUSED_B(20);
#undef DARTK_LAUNCH_FUSED_B
            DARTK_HIP(hipGetLastError());
        }
        if (useC) {
            ProfScope ps(DARTK_PROF_NE_SELF, st);
            rc = gn_launch_ne_self(d_cloud, nCloud, m, d_potential, slabsC, nBlocksC, st);
            if (rc) return rc;
        }
        ReduceArgs ra = {};
        ra.countStride = 2; ra.outStride = sysStride;
        int nTerms = 0;
        if (useA) ra.t[nTerms++] = ReduceTerm{slabsA, nBlocksA, S, cntA, 10, d_nValid, NE_BLOCK, sysA};
        if (useB) ra.t[nTerms++] = ReduceTerm{slabsB, nBlocksB, S, nullptr, 0, nullptr, 0, sysB};
        if (useC) ra.t[nTerms++] = ReduceTerm{slabsC, nBlocksC, Sa, nullptr, 0, nullptr, 0, sysC};
        SolveArgs a;
        a.sysA = useA ? sysA : nullptr; a.countA = cntA;
        a.sysB = useB ? sysB : nullptr; a.countB = cntB;
        a.sysC = useC ? sysC : nullptr;
        a.lambdaA = o->lambdaObsToMod; a.lambdaB = o->lambdaModToObs; a.lambdaC = o->lambdaIntersection;
        a.regularization = o->regularization; a.lmLambda = o->regularizationScaled;
        a.iterInfo = d_iterInfo ? d_iterInfo + 8 * it : nullptr;
        a.firstSystem = nullptr;
        a.sysStride = sysStride; a.countStride = 2; a.infoStride = 8 * o->numIterations;
        if (nTerms) {
            ProfScope ps(DARTK_PROF_REDUCE, st);
            hipLaunchKernelGGL(gn_reduce_kernel, dim3((S + 63) / 64, nTerms, nf), dim3(64 * GN_REDUCE_SEGS), 0, st, ra);
        }
        {
            ProfScope ps(DARTK_PROF_SOLVE, st);
            hipLaunchKernelGGL(gn_solve_single_kernel, dim3(nf), dim3(64), ldsS, st, m, a);
        }
        DARTK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gn_batch_pose_kernel, dim3(nf), dim3(64), 0, st, m, d_T_mcs, d_joints, 1);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

DARTK_COUNTER_ACCESS(gn)

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_profile_begin(void) {
    for (auto &e : g_prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    g_prof.clear();
    g_prof_on = true;
    return DARTK_OK;
}

int dartk_profile_end(float *h_ms_total, int32_t *h_launches) {
    DARTK_REQUIRE(h_ms_total && h_launches, "dartk_profile_end: null argument");
    g_prof_on = false;
    for (int c = 0; c < DARTK_PROF_CLASSES; ++c) { h_ms_total[c] = 0.0f; h_launches[c] = 0; }
    DARTK_HIP(hipDeviceSynchronize());
    for (auto &e : g_prof) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && e.cls >= 0 && e.cls < DARTK_PROF_CLASSES) {
            h_ms_total[e.cls] += ms;
            h_launches[e.cls] += 1;
        }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    g_prof.clear();
    return DARTK_OK;
}

int dartk_optimize_pose(dartk_model_t h, const float *d_vert, const float *d_norm, int width, int height,
                        const dartk_opt_options *o, const float *d_cloud, int nCloud, const int32_t *d_potential,
                        float *d_prediction, float *d_iterInfo, float *d_firstSystem, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_norm && o, "dartk_optimize_pose: null argument");
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    const bool useA = o->lambdaObsToMod > 0, useB = o->lambdaModToObs > 0;
    const bool useC = o->lambdaIntersection > 0 && nCloud > 0;
    if (useB) {
        DARTK_REQUIRE(d_prediction && o->predWidth > 0 && o->predHeight > 0, "mod->obs term needs a prediction buffer");
        DARTK_REQUIRE(mod->obsVoxels > 0, "mod->obs term needs a model with an observation SDF");
    }
    if (useC) DARTK_REQUIRE(d_cloud && d_potential, "intersection term needs a collision cloud and potential matrix");
    hipStream_t st = as_stream(stream);
    ModelView &m = mod->v;
    const int d = 6 + m.nJoints;
    const int S = sys_size(d);
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;
    const int nPix = width * height;
    const int tilesA = (nPix + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocksA = tilesA;                 // one tile of compacted valid pixels per workgroup (worst case: all valid)
    rc = ensure_partials(ws, (size_t)(2 * GN_MAX_BLOCKS + nBlocksA) * S);
    if (rc) return rc;
    float *slabsB = ws->partials, *slabsC = slabsB + (size_t)GN_MAX_BLOCKS * S, *slabsA = slabsC + (size_t)GN_MAX_BLOCKS * S;
    int *cntA = ws->counters, *cntB = ws->counters + 1, *d_nValid = ws->counters + 2;
    const int nVcBlocks = (nPix + VC_BLOCK - 1) / VC_BLOCK;
    rc = ensure_valid_idx(ws, (size_t)nPix + valid_pixel_work_ints(nPix, 1));
    if (rc) return rc;
    int *validIdx = ws->validIdx, *vcOffsets = ws->validIdx + nPix;
    float *d_T_gc = ws->sysScratch + 128;      // camera -> obs-SDF frame, frozen at the start (optimizer.cpp:446-448)
    float *sysA = ws->sysScratch + kMaxSys, *sysB = sysA + kMaxSys, *sysC = sysB + kMaxSys;   // reduced packed systems

    const int nPred = o->predWidth * o->predHeight;
    const int tilesB = (nPred + NE_BLOCK - 1) / NE_BLOCK;
    static const char *envB = getenv("DARTK_M2O_TILES_PER_WG");       // diagnostic: tiles per mod->obs workgroup
    const int tpw = (envB && atoi(envB) > 0) ? atoi(envB) : 1;
    const int wantB = (tilesB + tpw - 1) / tpw;
    const int nBlocksB = wantB < GN_MAX_BLOCKS ? wantB : GN_MAX_BLOCKS;
    const int tilesC = (nCloud + NE_BLOCK - 1) / NE_BLOCK;
    const int nBlocksC = tilesC < GN_MAX_BLOCKS ? tilesC : GN_MAX_BLOCKS;

    // The mod->obs chain (obs-SDF build, ray-cast, normal equations) and the obs->mod chain (valid-pixel compaction,
    // fused association + normal equations) only meet in the solve: with both terms on they run on two streams,
    // forked after the pose is final and joined before the reduction (the reference overlaps its ray-cast and its
    // association the same way, optimizer.cpp:610-639). Each chain alone leaves most of the chip idle. The longer
    // chain (ray-cast -> mod->obs) stays on the caller's stream with the reduction and the solve, so that the
    // iteration's critical path never waits on a cross-stream event; the shorter obs->mod chain takes the side stream
    // and has finished long before the join.
    const bool fork = useA && useB && o->numIterations > 0;
    // With both terms on, the trace and the obs->mod term share one launch on the caller's stream (trace_and_obs2mod_kernel):
    // the iteration then needs no fork and no join at all. The side stream is left for the self-intersection term, if any.
#ifdef DARTK_STAMPS
    const bool unionAB = false;         // the diagnostic build keeps the trace in its own kernel (its counters live there)
#else
    static const bool noUnion = getenv("DARTK_NO_UNION_LAUNCH") != nullptr;            // diagnostic: the two-stream form
    const bool unionAB = fork && !noUnion && m.nSdfs <= NE_BLOCK && m.nSdfs <= kMaxRaySdfs;
#endif
    const bool sideA = fork && (!unionAB || useC);
    hipStream_t stB = st, stA = sideA ? ws->auxStream : st, stO = fork ? ws->auxStream2 : st;
    // the two counters start at zero (later iterations: zeroed by the solve kernel): with the shared launch the call's first kernel
    // on `st`, the valid-pixel count, does it; else a memset before the fork
    const bool zeroInCount = unionAB;
    if (!zeroInCount) DARTK_HIP(hipMemsetAsync(ws->counters, 0, 2 * sizeof(int), st));
    // an error return after a fork must leave the caller's stream ordered after the side streams: they write the slab, index
    // and system buffers of this workspace, which the next call on `st` reuses
    struct JoinGuard {
        Workspace *w; hipStream_t st, a, o; bool on;
        ~JoinGuard() {
            if (!on) return;
            (void)hipEventRecord(w->evJoin, a); (void)hipEventRecord(w->evJoin2, o);
            (void)hipStreamWaitEvent(st, w->evJoin, 0); (void)hipStreamWaitEvent(st, w->evJoin2, 0);
        }
    } joinGuard{ws, st, stA, stO, false};
    if (fork) {
        DARTK_HIP(hipEventRecord(ws->evFork, st));
        if (sideA) DARTK_HIP(hipStreamWaitEvent(stA, ws->evFork, 0));
        DARTK_HIP(hipStreamWaitEvent(stO, ws->evFork, 0));
        joinGuard.on = true;
    }
    if (useB) {     // the obs SDF is first read by iteration 0's mod->obs kernel: built on a third stream beside the ray-cast
        {
            ProfScope ps(DARTK_PROF_OBS_SDF, stO);
            rc = gn_obs_sdf_build(d_vert, width, height, mod, o->focalLength, stO, d_T_gc);     // also freezes T_gc = T_mc (optimizer.cpp:446-448)
        }
        if (rc) return rc;
        if (fork) DARTK_HIP(hipEventRecord(ws->evObs, stO));
    }
    const size_t ldsA = sizeof(float) * (24 * (size_t)m.nFrames + (size_t)NE_BLOCK * row_len(d, false) + 8);
    const size_t ldsS = solve_lds_bytes(d, m.nFrames);
    DARTK_REQUIRE(ldsA <= 160 * 1024 && ldsS <= 160 * 1024, "LDS requirement exceeds 160 KiB");
    if (ldsS > 64 * 1024)
        DARTK_HIP(hipFuncSetAttribute((const void *)gn_solve_single_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsS));

    hipStream_t stV = unionAB ? st : stA;        // the valid-pixel list is read by the obs->mod tiles, wherever they run
    if (useA && o->numIterations > 0) {
        ProfScope ps(DARTK_PROF_OTHER, stV);
        unsigned long long *vcMasks = valid_pixel_masks(vcOffsets, nPix, 1);
        hipLaunchKernelGGL(valid_count_kernel, dim3(nVcBlocks), dim3(VC_BLOCK), 0, stV, (const float4 *)d_vert, nPix, vcOffsets,
                           zeroInCount ? ws->counters : nullptr, vcMasks);
        hipLaunchKernelGGL(valid_write_kernel, dim3(nVcBlocks), dim3(VC_BLOCK), 0, stV, (const unsigned long long *)vcMasks, nPix, vcOffsets, validIdx, d_nValid);
        DARTK_HIP(hipGetLastError());
    }
    const FusedObs2ModArgs fa = {(const float4 *)d_vert, (const float4 *)d_norm, validIdx, d_nValid, o->distThreshold, o->normThreshold,
                                 o->planeOffset, o->planeNormal[0], o->planeNormal[1], o->planeNormal[2], o->huberDelta, slabsA, cntA};
    const FrameBatch oneFrame{1, 0, 0, 0, 0, 0, 0, 0};
    TraceArgs ta = {};
    size_t ldsU = ldsA;
    if (unionAB) {
        rc = gn_raycast_reserve(ray_trace_floats(o->predWidth, o->predHeight, m.nSdfs));
        if (rc) return rc;
        // the tasks that survive the tile rectangles are known on the device only: a fixed number of workgroups walks them
        const long long tasks = (long long)((o->predWidth + 7) / 8) * ((o->predHeight + 7) / 8) * m.nSdfs;
        const long long want = (tasks + NE_BLOCK / 64 - 1) / (NE_BLOCK / 64);
        ta = TraceArgs{o->predFl[0], o->predFl[1], (float)(o->predWidth / 2), (float)(o->predHeight / 2), o->predWidth, o->predHeight,
                       (float2 *)ws->rayBuf, (int)(want < 2048 ? want : 2048), ws->sysScratch + 4 * kMaxSys};
        if (ldsU < sizeof(int) * kRectInts) ldsU = sizeof(int) * kRectInts;
        if (ldsU < sizeof(float) * (28 * (size_t)m.nSdfs + 1)) ldsU = sizeof(float) * (28 * (size_t)m.nSdfs + 1);
    }
    for (int it = 0; it < o->numIterations; ++it) {
        if (sideA && it > 0) {      // the pose of this iteration is final on `st` (previous solve)
            DARTK_HIP(hipEventRecord(ws->evFork, st));
            DARTK_HIP(hipStreamWaitEvent(stA, ws->evFork, 0));
        }
#define DARTK_LAUNCH_FUSED(PPT_)                                                                                \
    do {                                                                                                        \
        if (unionAB) {                                                                                          \
            auto kfn = trace_and_obs2mod_kernel<PPT_>;                                                          \
            if (ldsU > 64 * 1024)                                                                               \
                DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsU)); \
            hipLaunchKernelGGL(kfn, dim3(ta.nTraceWgs + nBlocksA + 1), dim3(NE_BLOCK), ldsU, st, ta, fa, m);    \
        } else {                                                                                                \
            auto kfn = fused_obs2mod_kernel<PPT_>;                                                              \
            if (ldsA > 64 * 1024)                                                                               \
                DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA)); \
            hipLaunchKernelGGL(kfn, dim3(nBlocksA), dim3(NE_BLOCK), ldsA, stA, fa, m, oneFrame);                \
        }                                                                                                       \
    } while (0)
        if (unionAB) {
            ProfScope ps(DARTK_PROF_RAYCAST, st);
            if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_FUSED(3);
            else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_FUSED(6);
            else DARTK_LAUNCH_FUSED(20);
            DARTK_HIP(hipGetLastError());
        }
        if (useB) {
            rc = gn_launch_raycast_ne_mod2obs(o->predFl[0], o->predFl[1], (float)(o->predWidth / 2), (float)(o->predHeight / 2),
                                              o->predWidth, o->predHeight, 0, m, d_prediction, d_T_gc, slabsB, nBlocksB, cntB, stB,
                                              (fork && it == 0) ? ws->evObs : nullptr, unionAB, unionAB ? ta.tables : nullptr);
            if (rc) return rc;
        }
        if (useA && !unionAB) {
            ProfScope ps(DARTK_PROF_FUSED_OBS2MOD, stA);
            if (S <= 3 * NE_BLOCK) DARTK_LAUNCH_FUSED(3);
            else if (S <= 6 * NE_BLOCK) DARTK_LAUNCH_FUSED(6);
            else DARTK_LAUNCH_FUSED(20);
            DARTK_HIP(hipGetLastError());
        }
#undef DARTK_LAUNCH_FUSED
        if (useC) {
            ProfScope ps(DARTK_PROF_NE_SELF, stA);
            rc = gn_launch_ne_self(d_cloud, nCloud, m, d_potential, slabsC, nBlocksC, stA);
            if (rc) return rc;
        }
        if (sideA) {
            DARTK_HIP(hipEventRecord(ws->evJoin, stA));
            DARTK_HIP(hipStreamWaitEvent(st, ws->evJoin, 0));
        }
        const int Sa = sys_size(d - 6);
        ReduceArgs ra = {};
        ra.countStride = 0; ra.outStride = 0;
        int nTerms = 0;
        if (useA) ra.t[nTerms++] = ReduceTerm{slabsA, nBlocksA, S, cntA, 10, d_nValid, NE_BLOCK, sysA};
        if (useB) ra.t[nTerms++] = ReduceTerm{slabsB, nBlocksB, S, nullptr, 0, nullptr, 0, sysB};
        if (useC) ra.t[nTerms++] = ReduceTerm{slabsC, nBlocksC, Sa, nullptr, 0, nullptr, 0, sysC};
        SolveArgs a;
        a.sysA = useA ? sysA : nullptr; a.countA = cntA;
        a.sysB = useB ? sysB : nullptr; a.countB = cntB;
        a.sysC = useC ? sysC : nullptr;
        a.lambdaA = o->lambdaObsToMod; a.lambdaB = o->lambdaModToObs; a.lambdaC = o->lambdaIntersection;
        a.regularization = o->regularization; a.lmLambda = o->regularizationScaled;
        a.iterInfo = d_iterInfo ? d_iterInfo + 8 * it : nullptr;
        a.firstSystem = (it == 0) ? d_firstSystem : nullptr;
        a.sysStride = a.countStride = a.infoStride = 0;
        if (nTerms) {
            ProfScope ps(DARTK_PROF_REDUCE, st);
            hipLaunchKernelGGL(gn_reduce_kernel, dim3((S + 63) / 64, nTerms), dim3(64 * GN_REDUCE_SEGS), 0, st, ra);
        }
        {
            ProfScope ps(DARTK_PROF_SOLVE, st);
            hipLaunchKernelGGL(gn_solve_single_kernel, dim3(1), dim3(64), ldsS, st, m, a);
        }
        DARTK_HIP(hipGetLastError());
    }
    joinGuard.on = false;       // every iteration joined stA before its reduction; stO joined through evObs in iteration 0
    return DARTK_OK;
}

int dartk_optimize_pose_batch(dartk_model_t h, int nFrames, const float *d_vert, const float *d_norm, int width, int height,
                              const dartk_opt_options *o, const float *d_cloud, int nCloud, const int32_t *d_potential,
                              float *d_T_mcs, float *d_joints, float *d_iterInfo, void *stream) {
    Model *mod = reinterpret_cast<Model *>(h);
    DARTK_REQUIRE(mod && d_vert && d_norm && o && d_T_mcs, "dartk_optimize_pose_batch: null argument");
    DARTK_REQUIRE(nFrames >= 1 && nFrames <= 65535, "nFrames %d out of range [1,65535]", nFrames);
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    DARTK_REQUIRE(mod->v.nJoints == 0 || d_joints, "dartk_optimize_pose_batch: joint array missing");
    const bool useB = o->lambdaModToObs > 0;
    const bool useC = o->lambdaIntersection > 0 && nCloud > 0;
    if (useB) {
        DARTK_REQUIRE(o->predWidth > 0 && o->predHeight > 0, "mod->obs term needs a prediction size");
        DARTK_REQUIRE(mod->obsVoxels > 0, "mod->obs term needs a model with an observation SDF");
        if (mod->v.nSdfs > NE_BLOCK || mod->v.nSdfs > kMaxRaySdfs) {
            set_error("dartk_optimize_pose_batch: the batched ray-cast replay needs at most %d SDFs, got %d", NE_BLOCK < kMaxRaySdfs ? NE_BLOCK : kMaxRaySdfs, mod->v.nSdfs);
            return DARTK_ERR_UNSUPPORTED;
        }
    }
    if (useC) DARTK_REQUIRE(d_cloud && d_potential, "intersection term needs a collision cloud and potential matrix");
    // frames in chunks that bound the per-call device memory (DARTK_BATCH_CHUNK, as dartk_optimize_poses_batch)
    const char *env = getenv("DARTK_BATCH_CHUNK");
    int chunk = env ? atoi(env) : 64;
    if (chunk < 1) chunk = 64;
    hipStream_t st = as_stream(stream);
    const size_t mapFloats = 4 * (size_t)width * height;
    for (int f0 = 0; f0 < nFrames; f0 += chunk) {
        const int nf = nFrames - f0 < chunk ? nFrames - f0 : chunk;
        int rc = optimize_pose_chunk(mod, nf, d_vert + f0 * mapFloats, d_norm + f0 * mapFloats, width, height, o, d_cloud, nCloud, d_potential,
                                     d_T_mcs + (size_t)f0 * 12, d_joints ? d_joints + (size_t)f0 * mod->v.nJoints : nullptr,
                                     d_iterInfo ? d_iterInfo + (size_t)f0 * o->numIterations * 8 : nullptr, st);
        if (rc) return rc;
    }
    return DARTK_OK;
}

}  // extern "C"
