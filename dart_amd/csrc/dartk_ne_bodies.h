// dartk_ne_bodies.h -- the bodies of the normal-equation kernels as device functions of a VIRTUAL workgroup index (bx of gx
// workgroups, row by): every kernel below is a thin __global__ wrapper around its body (dartk_obs2mod.hip, dartk_intersection.hip),
// and the single-frame multi-model loop runs several bodies side by side in ONE launch (union_terms_kernel, dartk_mod2obs.hip:
// the C2 recipe -- independent terms share a launch instead of a stream each, so an iteration is five launches on one stream).
// Generated once from the kernels by moving their text; the arithmetic is untouched.
#pragma once

#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"

namespace dartk {

#ifdef __HIPCC__

struct MultiArgs {
    int nModels;
    ModelView m[DARTK_MAX_MODELS];
    float distThr[DARTK_MAX_MODELS];
    float planeOffset[DARTK_MAX_MODELS];
    float pn[DARTK_MAX_MODELS][3];
    dartk_da_point *pts[DARTK_MAX_MODELS];
};

// words of LDS per model in the staged form: T_mc | T_fm[nFrames] | SDF -> frame | SDF headers
__host__ __device__ inline int assoc_stage_words(const ModelView &m) {
    return 12 + 12 * m.nFrames + m.nSdfs + (int)(sizeof(SdfDev) / 4) * m.nSdfs;
}

template <bool DBG, bool STAGED = false>
__device__ __forceinline__ void assoc_multi_body(const float4 *__restrict__ vert, int nPix, const MultiArgs *__restrict__ args,
                   int *__restrict__ lastElement, int *__restrict__ dbgDA, float *__restrict__ dbgErr, FrameBatch fb,
                   const int *__restrict__ validIdx, const int *__restrict__ nValid, int bx, int gx, int by) {
    extern __shared__ float lds[];
    const int frame = by;                    // frame of a batch (0 otherwise)
    vert += (size_t)frame * fb.vert;
    lastElement += (size_t)frame * fb.counts;
    if (STAGED) {
        int off = 0;
        const int nm = args->nModels;
        for (int mi = 0; mi < nm; ++mi) {
            const ModelView &m = args->m[mi];
            const size_t po = (size_t)frame * (size_t)m.poseStride;
            const int nT = 12 * m.nFrames, nS = m.nSdfs, nH = (int)(sizeof(SdfDev) / 4) * nS;
            float *p = lds + off;
            for (int i = threadIdx.x; i < 12 + nT; i += blockDim.x) p[i] = i < 12 ? m.T_mc[po + i] : m.T_fm[po + i - 12];
            int *pi = reinterpret_cast<int *>(p + 12 + nT);
            for (int i = threadIdx.x; i < nS; i += blockDim.x) pi[i] = m.sdfFrame[i];
            const int *src = reinterpret_cast<const int *>(m.sdfs);
            for (int i = threadIdx.x; i < nH; i += blockDim.x) pi[nS + i] = src[i];
            off += assoc_stage_words(m);
        }
        __syncthreads();
    }
    // with a list of the frame's valid pixels (gn_valid_pixels) the workgroups stride over the list instead of the map: the
    // launch then needs only as many workgroups as fill the chip, not one per 256 pixels of which most would find nothing
    const int nItems = validIdx ? nValid[frame] : nPix;
    for (int k0 = bx * blockDim.x; k0 < nItems; k0 += gx * blockDim.x) {
    const int k = k0 + threadIdx.x;
    const int index = validIdx ? (k < nItems ? validIdx[(size_t)frame * nPix + k] : nPix) : k;
    int aModel = -1, aGrid = -1;
    float sdfError = 1e20f;
    if (index < nPix) {
        if (DBG) {
            if (dbgDA) dbgDA[index] = -1;
            if (dbgErr) dbgErr[index] = __builtin_nanf("");
        }
        const float4 xc = vert[index];
        if (xc.w > 0) {
            const int nModels = args->nModels;
            int stageOff = 0;
            for (int mi = 0; mi < nModels; ++mi) {
                const ModelView &m = args->m[mi];
                const size_t po = (size_t)frame * (size_t)m.poseStride;
                const float *sM = lds + stageOff;                           // STAGED: this model's block
                stageOff += STAGED ? assoc_stage_words(m) : 0;
                const float *T_mc = STAGED ? sM : m.T_mc + po;
                const f3 xm = se3_apply_w(T_mc, xc.x, xc.y, xc.z, xc.w);
                f3 pn; pn.x = args->pn[mi][0]; pn.y = args->pn[mi][1]; pn.z = args->pn[mi][2];
                if (!(dot(xm, pn) >= args->planeOffset[mi])) continue;
                const float thr = args->distThr[mi];
                const int slot = keep_slot(thr);
                const float *T_fm = STAGED ? sM + 12 : m.T_fm + po;
                const int *sFrame = STAGED ? reinterpret_cast<const int *>(sM + 12 + 12 * m.nFrames) : m.sdfFrame;
                const SdfDev *sdfTab = STAGED ? reinterpret_cast<const SdfDev *>(sFrame + m.nSdfs) : m.sdfs;
                for (int g = 0; g < m.nSdfs; ++g) {
                    const int f = sFrame[g];
                    const f3 xf = se3_apply_w(T_fm + 12 * f, xm.x, xm.y, xm.z, xc.w);
                    const SdfDev &s = sdfTab[g];
                    const f3 xg = grid_coords(s, xf);
                    if (!in_bounds_grad_interp(s, xg) || !in_keep_box(s, slot, xg)) continue;
                    const float d = sdf_value<CNT_ASSOC>(m.grids, s, xg) * s.res;
                    if (d < sdfError) {
                        if (d * d < thr * thr) { sdfError = d; aGrid = g; aModel = mi; }
                    }
                }
            }
            if (DBG && aModel != -1) {
                if (dbgDA) dbgDA[index] = ((aModel << 16) | aGrid);
                if (dbgErr) dbgErr[index] = sdfError;
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int nModels = args->nModels;
    for (int mi = 0; mi < nModels; ++mi) {     // per-model wave-aggregated append
        const bool mine = (aModel == mi);
        const unsigned long long mask = __ballot(mine);
        if (!mask) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&lastElement[mi], __popcll(mask));
        base = __shfl(base, 0);
        if (mine) {
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            dartk_da_point p;
            p.index = index; p.dataAssociation = aGrid; p.error = sdfError;
            args->pts[mi][(size_t)frame * fb.pts + base + __popcll(mask & lt)] = p;
        }
    }
    }
}

template <int PPT, int RED>
__device__ __forceinline__ void ne_obs2mod_body(const dartk_da_point *__restrict__ pts, const float4 *__restrict__ vert, int nPoints, ModelView m_in,
                  float huberDelta, int redDims, const float *__restrict__ dtheta_dalpha,
                  const int *__restrict__ mapping, float *__restrict__ slabs, const int *__restrict__ d_count, FrameBatch fb, int bx, int gx, int by) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    select_frame(m, by);                        // frame of a batch (0 otherwise)
    pts += (size_t)by * fb.pts;
    vert += (size_t)by * fb.vert;
    if (d_count) nPoints = min(nPoints, d_count[(size_t)by * fb.counts]);      // device-resident loops: the list length stays on the device
    const int fullDims = 6 + m.nJoints;
    const int d = RED ? redDims : fullDims;
    const bool drop = (RED != 0);
    const int ld = row_len(d, drop);
    float *sT_fm = lds;
    float *sT_mf = sT_fm + 12 * m.nFrames;
    SdfDev *sSdf = reinterpret_cast<SdfDev *>(sT_mf + 12 * m.nFrames);
    float *sFull = reinterpret_cast<float *>(sSdf + m.nSdfs);          // [NE_BLOCK][fullDims|1] when RED
    const int fullLd = fullDims | 1;
    float *rows = sFull + (RED ? NE_BLOCK * fullLd : 0);
    stage_model(m, sT_fm, sT_mf, sSdf);
    float T_mc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T_mc[i] = m.T_mc[i];
    __syncthreads();

    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, drop);

    const int nTiles = (nPoints + NE_BLOCK - 1) / NE_BLOCK;
    for (int tile = bx; tile < nTiles; tile += gx) {
        const int k = tile * NE_BLOCK + threadIdx.x;
        float *row = rows + threadIdx.x * ld;
        if (k < nPoints) {
            const dartk_da_point pt = pts[k];
            const float4 xc = vert[pt.index];
            const f3 xm = se3_apply_w(T_mc, xc.x, xc.y, xc.z, xc.w);
            const int g = pt.dataAssociation;
            const int frame = m.sdfFrame[g];
            const f3 xf = se3_apply_w(sT_fm + 12 * frame, xm.x, xm.y, xm.z, xc.w);
            const SdfDev s = sSdf[g];
            const f3 xg = grid_coords(s, xf);
            const f3 gf = sdf_gradient<CNT_NE_O2M>(m.grids, s, xg);
            const f3 gm = se3_rotate(sT_mf + 12 * frame, gf);
            float *Jfull = RED ? (sFull + threadIdx.x * fullLd) : row;
            error_jacobian<false>(Jfull, xm, xc.w, m.depMask[frame], gm, m.nJoints, m.jointType, m.jointAxis, sT_fm, sT_mf);
            if (RED) reduce_jacobian<RED>(row, Jfull, 6, fullDims, redDims, dtheta_dalpha, mapping);
            const float err = pt.error;
            const bool inlier = fabsf(err) < huberDelta;
            float r, E;
            if (inlier) { r = -err; E = (float)(0.5 * (double)r * (double)r); }
            else {
                const float v = (err < 0) ? -huberDelta : huberDelta;
                r = -v;
                E = (float)((double)huberDelta * ((double)fabsf(err) - 0.5 * (double)huberDelta));
            }
            row[d] = r; row[d + 1] = E; row[d + 2] = 1.0f;
            if (drop) {
                for (int i = 0; i < d; ++i) row[d + 3 + i] = inlier ? row[i] : 0.0f;
            }
        }
        __syncthreads();
        const int n = min(NE_BLOCK, nPoints - tile * NE_BLOCK);
        acc.accumulate(rows, ld, n);
        __syncthreads();
    }
    acc.store(slabs + ((size_t)by * gx + bx) * sys_size(d), sys_size(d));
}

template <int PPT, int RED>
__device__ __forceinline__ void ne_self_intersection_body(const float4 *__restrict__ sites, int nSites, ModelView m_in, const int *__restrict__ pot,
                            int redDims, const float *__restrict__ dtheta_dalpha, const int *__restrict__ mapping,
                            float *__restrict__ slabs, int bx, int gx, int by) {
    extern __shared__ float lds[];
    ModelView m = m_in;
    select_frame(m, by);                    // frame of a batch (0 otherwise)
    const int fullDims = 6 + m.nJoints;
    const int d = (RED ? redDims : fullDims) - 6;
    const int ld = row_len(d, false);
    float *sT_fm = lds;
    float *sT_mf = sT_fm + 12 * m.nFrames;
    const int fullLd = (fullDims - 6) | 1;
    float *sFull = sT_mf + 12 * m.nFrames;
    float *rows = sFull + (RED ? NE_BLOCK * fullLd : 0);
    int *waveCnt = reinterpret_cast<int *>(rows + NE_BLOCK * ld);
    stage_model(m, sT_fm, sT_mf, nullptr);
    __syncthreads();
    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, false);
    const int nTiles = (nSites + NE_BLOCK - 1) / NE_BLOCK;
    for (int tile = bx; tile < nTiles; tile += gx) {
        const int index = tile * NE_BLOCK + threadIdx.x;
        Collision c; c.hit = false;
        f3 vm; vm.x = vm.y = vm.z = 0.f;
        int srcFrame = 0;
        if (index < nSites) {
            const float4 v = sites[index];
            const int srcGrid = (int)roundf(v.w);
            srcFrame = m.sdfFrame[srcGrid];
            vm = se3_apply_w(sT_mf + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
            c = first_collision(m, sT_fm, vm, pot, srcGrid);
        }
        int nActive;
        const int slot = compact_slot<NE_BLOCK>(c.hit, waveCnt, nActive);
        if (c.hit) {
            float *row = rows + slot * ld;
            const SdfDev &s = m.sdfs[c.dstGrid];
            const int dstFrame = m.sdfFrame[c.dstGrid];
            const f3 gf = sdf_gradient<CNT_INTER>(m.grids, s, c.vg);
            const f3 gm = se3_rotate(sT_mf + 12 * dstFrame, gf);
            float *Jfull = RED ? (sFull + threadIdx.x * fullLd) : row;
            error_jacobian<true>(Jfull, vm, 1.0f, m.depMask[srcFrame], gm, m.nJoints, m.jointType, m.jointAxis, sT_fm, sT_mf);
            if (RED) reduce_jacobian<RED>(row, Jfull, 0, fullDims, redDims, dtheta_dalpha, mapping);
            row[d] = c.residual;
            row[d + 1] = (float)(0.5 * (double)c.residual * (double)c.residual);
            row[d + 2] = 1.0f;
        }
        __syncthreads();
        if (nActive) acc.accumulate(rows, ld, nActive);
        __syncthreads();
    }
    acc.store(slabs + ((size_t)by * gx + bx) * sys_size(d), sys_size(d));
}

template <int PPT, int RED>
__device__ __forceinline__ void ne_intersection_body(const float4 *__restrict__ sites, int nSites, ModelView src_in, ModelView dst_in,
                       const float *__restrict__ T_ds_sd /* [24] */, int redDims, const float *__restrict__ dtheta_dalpha,
                       const int *__restrict__ mapping, float *__restrict__ slabs, int tabStride, int bx, int gx, int by) {
    extern __shared__ float lds[];
    ModelView src = src_in, dst = dst_in;
    select_frame(src, by);                  // frame of a batch (0 otherwise)
    select_frame(dst, by);
    T_ds_sd += (size_t)by * tabStride;
    const int fullDims = 6 + src.nJoints;
    const int d = RED ? redDims : fullDims;
    const int ld = row_len(d, false);
    float *sT_fm = lds;                               // source transforms
    float *sT_mf = sT_fm + 12 * src.nFrames;
    float *dT_fm = sT_mf + 12 * src.nFrames;          // destination transforms
    float *dT_mf = dT_fm + 12 * dst.nFrames;
    const int fullLd = fullDims | 1;
    float *sFull = dT_mf + 12 * dst.nFrames;
    float *rows = sFull + (RED ? NE_BLOCK * fullLd : 0);
    int *waveCnt = reinterpret_cast<int *>(rows + NE_BLOCK * ld);
    stage_model(src, sT_fm, sT_mf, nullptr);
    stage_model(dst, dT_fm, dT_mf, nullptr);
    float T_ds[12], T_sd[12];
    if (T_ds_sd) {
#pragma unroll
        for (int i = 0; i < 12; ++i) { T_ds[i] = T_ds_sd[i]; T_sd[i] = T_ds_sd[12 + i]; }
    } else {        // no table: T_ds = T_mc(dst) * T_cm(src) and its inverse here, the arithmetic of pair_transform_kernel (optimizer.cpp:396-397)
        float A[12], B[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { A[i] = dst.T_mc[i]; B[i] = src.T_cm[i]; }
        se3_mul(A, B, T_ds);
        se3_invert(T_ds, T_sd);
    }
    __syncthreads();
    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, false);
    const int nTiles = (nSites + NE_BLOCK - 1) / NE_BLOCK;
    for (int tile = bx; tile < nTiles; tile += gx) {
        const int index = tile * NE_BLOCK + threadIdx.x;
        Collision c; c.hit = false;
        f3 vsm; vsm.x = vsm.y = vsm.z = 0.f;
        int srcFrame = 0;
        if (index < nSites) {
            const float4 v = sites[index];
            const int srcGrid = (int)roundf(v.w);
            srcFrame = src.sdfFrame[srcGrid];
            vsm = se3_apply_w(sT_mf + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
            const f3 vdm = se3_apply_w(T_ds, vsm.x, vsm.y, vsm.z, 1.0f);
            c = first_collision(dst, dT_fm, vdm, nullptr, 0);
        }
        int nActive;
        const int slot = compact_slot<NE_BLOCK>(c.hit, waveCnt, nActive);
        if (c.hit) {
            float *row = rows + slot * ld;
            const SdfDev &s = dst.sdfs[c.dstGrid];
            const int dstFrame = dst.sdfFrame[c.dstGrid];
            const f3 gf = sdf_gradient<CNT_INTER>(dst.grids, s, c.vg);
            const f3 gdm = se3_rotate(dT_mf + 12 * dstFrame, gf);
            const f3 gsm = se3_rotate(T_sd, gdm);
            float *Jfull = RED ? (sFull + threadIdx.x * fullLd) : row;
            error_jacobian<false>(Jfull, vsm, 1.0f, src.depMask[srcFrame], gsm, src.nJoints, src.jointType, src.jointAxis, sT_fm, sT_mf);
            if (RED) reduce_jacobian<RED>(row, Jfull, 6, fullDims, redDims, dtheta_dalpha, mapping);
            row[d] = c.residual;
            row[d + 1] = (float)(0.5 * (double)c.residual * (double)c.residual);
            row[d + 2] = 1.0f;
        }
        __syncthreads();
        if (nActive) acc.accumulate(rows, ld, nActive);
        __syncthreads();
    }
    acc.store(slabs + ((size_t)by * gx + bx) * sys_size(d), sys_size(d));
}

#endif  // __HIPCC__

}  // namespace dartk
