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
#include <type_traits>

#include "dartk_internal.h"

namespace dartk {

#ifdef __HIPCC__

__host__ __device__ constexpr int jtj_size(int d) { return (d * (d + 1)) >> 1; }
__host__ __device__ constexpr int sys_size(int d) { return d + jtj_size(d) + 1; }
// LDS row length (floats), padded to an odd stride so that consecutive rows start on different banks
__host__ __device__ inline int row_len(int d, bool drop) {
    int n = (drop ? 2 * d + 3 : d + 3);
    return n | 1;
}

// weighted rows (contact priors): [ J | r | E | 1 | w ], J^T J entry (i,j), j <= i, accumulates (w*J_j)*J_i
__host__ __device__ inline int row_len_w(int d) { return (d + 4) | 1; }

template <int BLOCK, int PPT>
struct PairAcc {
    float acc[PPT];
    int ia[PPT], ib[PPT];

    // drop: J^T J from the second copy of the row (both factors); weighted: (w*J_j)*J_i with the weight read from the
    // row (ContactPrior's w*J*J^T, contact_prior.cpp:129) -- use accumulate_w() then
    __device__ __forceinline__ void init(int d, bool drop, bool weighted = false) {
        const int S = sys_size(d);
        const int base = drop ? d + 3 : 0;
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
                if (weighted) { a = j; b = i; }
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

    // weighted rows: J^T J slots multiply by the row's weight first, every other slot by the constant 1 of the row
    __device__ __forceinline__ void accumulate_w(const float *rows, int ld, int n, int d) {
        int iw[PPT];
        const int S = sys_size(d);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int o = (int)threadIdx.x + k * BLOCK;
            iw[k] = (o >= d && o < S - 1) ? d + 3 : d + 2;
        }
        for (int p = 0; p < n; ++p) {
            const float *r0 = rows + p * ld;
#pragma unroll
            for (int k = 0; k < PPT; ++k) acc[k] = __builtin_fmaf(r0[iw[k]] * r0[ia[k]], r0[ib[k]], acc[k]);
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

// Register-tiled variant for the batched (NoObs) terms, where the row reduction is the LDS-bound part: a row holds
//   R = [ r | E | Jw_0 .. Jw_{d-1} | .. ]      L = [ J_0 .. J_{d-1} | 1 | 0.. ]
// with r the effective residual and Jw the (weighted) Jacobian. Unweighted rows (Jw == J) store one array
// [ r | E | J | 1 | 0.. ] and L aliases it at offset 2; weighted rows store L after R (offset tile_pad(d+2)).
// The packed system is read off the outer product
// M[i][c] = sum_rows L_i * R_c:  J^T e_i = M[i][0],  e = M[d][1],  J^T J (i, j<=i) = M[i][2+j]. A thread owns 4x4 tiles of
// M (two 16-byte LDS reads feed 16 FMAs); only tiles that hold a wanted entry are enumerated. Every entry is still ONE
// fma chain over the rows in their order, so the sums are bit-identical to PairAcc's.
__host__ __device__ inline int tile_pad(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int tile_l_off(int d, bool weighted) { return weighted ? tile_pad(d + 2) : 2; }
__host__ __device__ inline int tile_row_len(int d, bool weighted) { return tile_pad(tile_l_off(d, weighted) + tile_pad(d + 1)); }
__host__ __device__ inline int tile_jmax(int I, int d) {        // last wanted column tile of row tile I
    const int imax = (4 * I + 3 < d - 1) ? 4 * I + 3 : d - 1;
    return (imax + 2) >> 2;
}
__host__ __device__ inline int tile_count(int d) {
    int n = 0;
    for (int I = 0; 4 * I < d + 1; ++I) n += tile_jmax(I, d) + 1;
    return n;
}

template <int BLOCK, int TPT>
struct TileAcc {
    float acc[TPT][16];
    int I[TPT], J[TPT];       // tile coordinates, I < 0: idle

    __device__ __forceinline__ void init(int d) {
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            int t = (int)threadIdx.x + k * BLOCK;
            I[k] = -1; J[k] = 0;
            for (int i = 0; 4 * i < d + 1; ++i) {
                const int n = tile_jmax(i, d) + 1;
                if (t < n) { I[k] = i; J[k] = t; break; }
                t -= n;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
        }
    }

    // rows: LDS, 16-byte aligned, n rows of stride ld (a multiple of 4); lOff = tile_l_off() (2: 8-byte aligned L quads)
    template <bool WEIGHTED>
    __device__ __forceinline__ void load_row(const float *pl, const float *pr, float *lv, float *rv) const {
        if (WEIGHTED) {
            const float4 l = *reinterpret_cast<const float4 *>(pl);
            lv[0] = l.x; lv[1] = l.y; lv[2] = l.z; lv[3] = l.w;
        } else {
            const float2 l0 = *reinterpret_cast<const float2 *>(pl), l1 = *reinterpret_cast<const float2 *>(pl + 2);
            lv[0] = l0.x; lv[1] = l0.y; lv[2] = l1.x; lv[3] = l1.y;
        }
        const float4 r = *reinterpret_cast<const float4 *>(pr);
        rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
    }
    __device__ __forceinline__ void fma_row(int k, const float *lv, const float *rv) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[k][4 * a + b] = __builtin_fmaf(lv[a], rv[b], acc[k][4 * a + b]);
    }
    template <bool WEIGHTED>
    __device__ __forceinline__ void accumulate(const float *rows, int ld, int n, int lOff) {
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            if (I[k] < 0) continue;
            const float *pl = rows + lOff + 4 * I[k], *pr = rows + 4 * J[k];
            int p = 0;
            for (; p + 4 <= n; p += 4, pl += 4 * ld, pr += 4 * ld) {       // four rows of loads in flight, rows added in order
                float lv[4][4], rv[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<WEIGHTED>(pl + u * ld, pr + u * ld, lv[u], rv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) fma_row(k, lv[u], rv[u]);
            }
            for (; p < n; ++p, pl += ld, pr += ld) {
                float lv[4], rv[4];
                load_row<WEIGHTED>(pl, pr, lv, rv);
                fma_row(k, lv, rv);
            }
        }
    }

    // add v to the e slot (by the thread that owns it)
    __device__ __forceinline__ void add_e(int d, double v) {
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            if (I[k] == (d >> 2) && J[k] == 0) {
                const int e = 4 * (d & 3) + 1;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q == e) acc[k][q] = (float)((double)acc[k][q] + v);
            }
    }

    __device__ __forceinline__ void store(float *slab, int d) const {
        const int S = sys_size(d);
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            if (I[k] < 0) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * I[k] + a, c = 4 * J[k] + b;
                    int o = -1;
                    if (i < d && c == 0) o = i;
                    else if (i == d && c == 1) o = S - 1;
                    else if (i < d && c >= 2 && c - 2 <= i) o = d + ((i * (i + 1)) >> 1) + (c - 2);
                    if (o >= 0) slab[o] = acc[k][4 * a + b];
                }
        }
    }
};

// The same outer product on the matrix cores, for systems of at most 32 dimensions (the ContactGrasp block: hand 26 + object 6; every
// intersection term of the hands here). rocprof on the batched term kernels showed the contraction -- not the SDF gathers -- to be
// the largest single part of the contact-prior kernel (37 % of its time on the VALU, profiles/r03_*): north_star's condition for
// moving it to MFMA. v_mfma_f32_32x32x2_f32 takes A[i][k] = L_i of row k and B[k][c] = R_c of row k, two rows per instruction, and
// is an exact fp32 fma chain over k (MI355X_MICROARCH.md, "exact f32 = fmaf chain, bitwise"): with the rows fed in their order every
// entry M[i][c], i < 32, c < 32, is the SAME chain TileAcc forms, bit for bit. One wave issues one instruction per two rows instead
// of ~36 (reads + FMAs of its 4x4 tiles). What a 32 x 32 block cannot hold -- J^T J columns 30, 31 (c = 32, 33) and, for d = 32, the e
// slot (i = 32) -- are at most four chains, kept by four lanes of the workgroup's second wave. BLOCK must be 128.
// LDSC: the 32 x 32 block lives in LDS between accumulate() calls (register e of lane l at cLds[64 e + l]; only the registers that
// hold a row i <= d, mfma_lds_floats(d) floats -- A is zero for i > d, so the others stay exactly zero) and
// is in registers only while the chain runs -- for kernels whose other phases need the 16 registers (the intersection terms: rows
// are rare there and the test loop spilled to scratch with the block resident). Same instructions on the same values in the same
// order: bit-identical to the register-resident form.
__host__ __device__ inline int mfma_lds_regs(int d) { return 4 * (d / 8 + 1) < 16 ? 4 * (d / 8 + 1) : 16; }     // rows 8g .. 8g+7 sit in registers 4g .. 4g+3
__host__ __device__ inline int mfma_lds_floats(int d) { return 64 * mfma_lds_regs(d); }
template <int BLOCK, bool LDSC = false>
struct MfmaAcc {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    struct Empty {};
    std::conditional_t<LDSC, Empty, f32x16> c;
    float *cLds;            // LDSC only
    float extra;            // leftover chain of this thread (second wave, lanes 0..3)
    int ei, ec;             // its operands: L index, R index; ei < 0: none

    __device__ __forceinline__ void init(int d, int tid = (int)threadIdx.x) {
        if constexpr (LDSC) {
            if (tid < 64) {
#pragma unroll
                for (int e = 0; e < 16; ++e) if (e < mfma_lds_regs(d)) cLds[64 * e + tid] = 0.0f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = 0.0f;
        }
        extra = 0.0f; ei = -1; ec = 0;
        const int q = tid - 64;
        if (q >= 0 && q < 4) {
            // wanted entries outside the block: (i, j) with j in {30, 31}, j <= i < d  ->  c = j + 2;   e (i = d, c = 1) when d == 32
            int k = 0;
            for (int j = 30; j < 32 && j < d; ++j)
                for (int i = j; i < d && i < 32; ++i) { if (k == q) { ei = i; ec = j + 2; } ++k; }
            if (d == 32) { if (k == q) { ei = 32; ec = 1; } ++k; }
        }
    }

    // rows: LDS, n rows of stride ld; L_i = rows[lOff + i] (i <= d), R_c = rows[c]
    template <bool WEIGHTED>
    __device__ __forceinline__ void accumulate(const float *rows, int ld, int n, int lOff) {
        const int tid = (int)threadIdx.x;
        if (tid < 64) {
            const int half = tid >> 5, col = tid & 31;
            const int dTop = dims;
            const float *pa = rows + half * ld + lOff + col, *pb = rows + half * ld + col;
            f32x16 acc;
            if constexpr (LDSC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = (e < mfma_lds_regs(dTop)) ? cLds[64 * e + tid] : 0.0f;
            } else {
                acc = c;
            }
            for (int p = 0; p < n; p += 2, pa += 2 * ld, pb += 2 * ld) {
                float a = 0.0f, b = 0.0f;
                if (p + half < n) { a = (col <= dTop) ? *pa : 0.0f; b = *pb; }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if constexpr (LDSC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) if (e < mfma_lds_regs(dTop)) cLds[64 * e + tid] = acc[e];
            } else {
                c = acc;
            }
        } else if (ei >= 0) {
            const float *pl = rows + lOff + ei, *pr = rows + ec;
            for (int p = 0; p < n; ++p, pl += ld, pr += ld) extra = __builtin_fmaf(*pl, *pr, extra);
        }
    }

    __device__ __forceinline__ void add_e(int d, double v) {
        const int tid = (int)threadIdx.x;
        if (d == 32) {
            if (tid >= 64 && ei == 32) extra = (float)((double)extra + v);
            return;
        }
        if (tid < 64 && (tid & 31) == 1) {          // M[d][1]: column 1, row d = (reg & 3) + 8 (reg >> 2) + 4 half
            const int half = tid >> 5;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((e & 3) + 8 * (e >> 2) + 4 * half == d) {
                    if constexpr (LDSC) cLds[64 * e + tid] = (float)((double)cLds[64 * e + tid] + v);
                    else c[e] = (float)((double)c[e] + v);
                }
        }
    }

    // (LDSC: every accumulate() ends in a workgroup barrier of the caller's before store() runs; wave 0 reads its own entries anyway)
    __device__ __forceinline__ void store(float *slab, int d) const {
        const int S = sys_size(d);
        const int tid = (int)threadIdx.x;
        if (tid < 64) {
            const int half = tid >> 5, cc = tid & 31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                int o = -1;
                if (i < d && cc == 0) o = i;
                else if (i == d && cc == 1) o = S - 1;
                else if (i < d && cc >= 2 && cc - 2 <= i) o = d + ((i * (i + 1)) >> 1) + (cc - 2);
                if (o >= 0) {
                    if constexpr (LDSC) slab[o] = cLds[64 * e + tid];
                    else slab[o] = c[e];
                }
            }
        } else if (ei >= 0) {
            slab[ei == d ? S - 1 : d + ((ei * (ei + 1)) >> 1) + (ec - 2)] = extra;
        }
    }

    int dims;       // set by the kernel before the first accumulate (rows beyond L_d hold another row's data)
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

// result[o] = sum_b slabs[b][o] in a fixed order with fp64 accumulation. keep==0 writes zeros (the reference's
// "nElements > 10" gate leaves the memset system untouched); the gated form reads that decision on the device
// (*d_count > minCount).
int launch_reduce_slabs(const float *slabs, int nBlocks, int S, float *result, int keep, hipStream_t stream);
int launch_reduce_slabs_direct(const float *slabs, int nBlocks, int S, float *result, hipStream_t stream);
int launch_reduce_slabs_gated(const float *slabs, int nBlocks, int S, float *result, const int *d_count, int minCount,
                              hipStream_t stream);

#endif  // __HIPCC__

}  // namespace dartk
