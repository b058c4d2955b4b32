// dartk_reduce.h -- deterministic normal-equation reduction (replaces computeSquaredLossResult's float
// atomics, kernel_common.h:141-153).
//
// Every workgroup writes one LDS row per active point:  [ J_0 .. J_{d-1} | r | E | 1 | (Jw_0 .. Jw_{d-1}) ]
// where r is the effective residual (J^T e += J*r), E the point's loss value (e += E*1) and Jw the Jacobian
// used for J^T J (only present in the DROP layout, where Huber outliers keep J^T e but drop J^T J,
// obsToMod.cu:404-415). The packed system has S = (d+1)(d+2)/2 entries, each the sum over points of ONE
// product of two row entries, so output slot o is owned by one thread which walks the rows with two
// conflict-free LDS reads and one FMA per point. Partial systems stay in registers across the tiles a
// workgroup processes, are written once per workgroup to a [nBlocks][S] slab, and a second tiny kernel adds
// the slabs in fp64 in a fixed order: bitwise reproducible, no float atomics.
#pragma once

#include "dartk_internal.h"

namespace dartk {

#ifdef __HIPCC__

__host__ __device__ inline int jtj_size(int d) { return (d * (d + 1)) >> 1; }
__host__ __device__ inline int sys_size(int d) { return d + jtj_size(d) + 1; }
// LDS row length (floats), padded to an odd stride so that consecutive rows start on different banks
__host__ __device__ inline int row_len(int d, bool drop) {
    int n = (drop ? 2 * d + 3 : d + 3);
    return n | 1;
}

template <int BLOCK, int PPT>
struct PairAcc {
    float acc[PPT];
    int ia[PPT], ib[PPT];

    // drop: J^T J from the second copy of the row (both factors); weighted: first factor from the second copy
    // (w*J), second from the plain row (ContactPrior's w*J*J^T, contact_prior.cpp:129)
    __device__ __forceinline__ void init(int d, bool drop, bool weighted = false) {
        const int S = sys_size(d);
        const int base = (drop || weighted) ? d + 3 : 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int o = (int)threadIdx.x + k * BLOCK;
            acc[k] = 0.0f;
            int a = d + 2, b = d + 2;                 // idle slots multiply 1*1 into an unused accumulator
            if (o < d) { a = o; b = d; }              // J^T e
            else if (o == S - 1) { a = d + 1; b = d + 2; }   // e
            else if (o < S) {
                const int t = o - d;
                int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while (((i * (i + 1)) >> 1) > t) --i;
                while ((((i + 1) * (i + 2)) >> 1) <= t) ++i;
                const int j = t - ((i * (i + 1)) >> 1);
                if (weighted) { a = base + j; b = i; }
                else { a = base + i; b = base + j; }
            }
            ia[k] = a; ib[k] = b;
        }
    }

    // rows: LDS, n rows of stride ld
    __device__ __forceinline__ void accumulate(const float *rows, int ld, int n) {
        int p = 0;
        for (; p + 4 <= n; p += 4) {
            const float *r0 = rows + p * ld;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                float s = acc[k];
                s = __builtin_fmaf(r0[ia[k]], r0[ib[k]], s);
                s = __builtin_fmaf(r0[ld + ia[k]], r0[ld + ib[k]], s);
                s = __builtin_fmaf(r0[2 * ld + ia[k]], r0[2 * ld + ib[k]], s);
                s = __builtin_fmaf(r0[3 * ld + ia[k]], r0[3 * ld + ib[k]], s);
                acc[k] = s;
            }
        }
        for (; p < n; ++p) {
            const float *r0 = rows + p * ld;
#pragma unroll
            for (int k = 0; k < PPT; ++k) acc[k] = __builtin_fmaf(r0[ia[k]], r0[ib[k]], acc[k]);
        }
    }

    // add v to output slot o (by the thread that owns it)
    __device__ __forceinline__ void add_at(int o, double v) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if ((int)threadIdx.x + k * BLOCK == o) acc[k] = (float)((double)acc[k] + v);
    }

    __device__ __forceinline__ void store(float *slab, int S) const {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int o = (int)threadIdx.x + k * BLOCK;
            if (o < S) slab[o] = acc[k];
        }
    }
};

// Deterministic compaction slot for this thread inside its workgroup: returns the row index (or -1 when
// inactive) and the workgroup's active count. `waveCnt` is an LDS array of BLOCK/64 ints.
template <int BLOCK>
__device__ __forceinline__ int compact_slot(bool active, int *waveCnt, int &total) {
    const unsigned long long mask = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) waveCnt[wave] = __popcll(mask);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const int c = waveCnt[w];
        if (w < wave) off += c;
        tot += c;
    }
    total = tot;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    return active ? off + __popcll(mask & lt) : -1;
}

// result[o] = scale * sum_b slabs[b][o], fixed order, fp64 accumulation. keep==0 writes zeros (the
// reference's "nElements > 10" gate leaves the memset system untouched).
__global__ void reduce_slabs_kernel(const float *__restrict__ slabs, int nBlocks, int S, float *__restrict__ result,
                                    int keep);

int launch_reduce_slabs(const float *slabs, int nBlocks, int S, float *result, int keep, hipStream_t stream);

#endif  // __HIPCC__

}  // namespace dartk
