// dartk_noobs.hip -- ContactGrasp refinement: OptimizerNoObs::optimizePoses / getError (optimizer.cpp:783-1076),
// ContactPrior::computeContribution (contact_prior.cpp:8-205, CPU-only in the reference) and the batch of
// independent grasp candidates (BASELINE configs 3/4), device-resident.
//
// The reference evaluates ~2500 contact priors one by one on the host (host SDF reads, sparse coeffRef inserts into a
// 7.5k-dimensional system whose prior rows are identically zero, SURVEY A-7) around three blocking kernel round trips
// per iteration. Here one iteration of ALL candidates of a batch is two launches:
//   noobs_terms_kernel  grid (tiles of every term, candidate): self-intersection, pairwise intersection and contact
//                       priors, each tile -> one partial packed system (deterministic slabs, no float atomics)
//   noobs_solve_kernel  grid (candidate), one wave: fixed-order reduction of the candidate's slabs, assembly in the
//                       reference's order, regularisation + LM damping, the reference's upper-triangular back
//                       substitution (SURVEY A-1), pose update with joint clamps, forward kinematics
// Poses of the batch live in one HBM array; SDFs, clouds, priors and kinematic tables are shared by all candidates.
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_fk.h"

#include <string.h>
#include <vector>

namespace dartk {

enum { TERM_SELF = 0, TERM_INTER = 1, TERM_PRIORS = 2 };
static constexpr int NOOBS_MAX_TERMS = DARTK_MAX_MODELS * DARTK_MAX_MODELS + 1;

struct NoObsTerm {
    int type, src, dst;
    int nPoints, nBlocks;        // points (cloud sites / priors) and workgroups (= slabs) of this term per candidate
    int blockStart;              // first blockIdx.x of the term
    int dims, S;                 // dimension / packed length of the term's system
    int slabOff;                 // offset (floats) of the term's slabs inside a candidate's slab area
    const float4 *cloud;
    const int *pot;
};

struct NoObsProblem {            // device-visible description shared by both kernels (lives in HBM)
    int nModels, nTerms, D, nPriors;
    int poseStride;              // floats per candidate in the pose array
    int slabStride;              // floats per candidate in the slab array
    int jointStride;             // sum of joints over models
    int off[DARTK_MAX_MODELS];       // offset of model m in the global system
    int dims[DARTK_MAX_MODELS];
    int poseOff[DARTK_MAX_MODELS];   // offset of model m's pose block inside a candidate's pose record
    int jointOff[DARTK_MAX_MODELS];
    ModelView model[DARTK_MAX_MODELS];   // static tables only; pose pointers are rebased per candidate
    NoObsTerm term[NOOBS_MAX_TERMS];
    const dartk_contact_prior *priors;
};

// pose block of one model of one candidate: T_mc | T_cm | joints | T_mf | T_fm
__device__ __forceinline__ ModelView rebase(const ModelView &s, float *pose) {
    ModelView v = s;
    v.T_mc = pose; v.T_cm = pose + 12; v.joints = pose + 24;
    const int nj = s.nJoints > 0 ? s.nJoints : 1;
    v.T_mf = pose + 24 + nj; v.T_fm = v.T_mf + 12 * s.nFrames;
    return v;
}
static int pose_block_floats(const ModelView &m) { return 24 + (m.nJoints > 0 ? m.nJoints : 1) + 24 * m.nFrames; }

// import poses (T_mc, joints) of the batch into the pose array and run forward kinematics; grid (model, candidate)
__global__ void __launch_bounds__(64)
noobs_load_poses_kernel(const NoObsProblem *__restrict__ P, float *__restrict__ poses, const float *__restrict__ T_mcs,
                        const float *__restrict__ joints) {
    __shared__ float sT[12 * (DARTK_MAX_JOINTS + 1)];
    const int mi = blockIdx.x, c = blockIdx.y;
    float *pose = poses + (size_t)c * P->poseStride + P->poseOff[mi];
    const ModelView m = rebase(P->model[mi], pose);
    const int lane = threadIdx.x;
    if (lane < 12) pose[lane] = T_mcs[((size_t)c * P->nModels + mi) * 12 + lane];
    if (lane < m.nJoints) m.joints[lane] = joints[(size_t)c * P->jointStride + P->jointOff[mi] + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        float T[12], inv[12];
        for (int i = 0; i < 12; ++i) T[i] = T_mcs[((size_t)c * P->nModels + mi) * 12 + i];
        se3_invert(T, inv);
        for (int i = 0; i < 12; ++i) m.T_cm[i] = inv[i];
    }
    fk_wave(m, 0, sT);
}

// -----------------------------------------------------------------------------------------------------
// all terms of one iteration. LDS: per model [T_fm | T_mf] (24*nFrames) | rows | wave counters | scalars
// -----------------------------------------------------------------------------------------------------
template <int PPT>
__global__ void __launch_bounds__(NE_BLOCK)
noobs_terms_kernel(const NoObsProblem *__restrict__ P, float *__restrict__ poses, float *__restrict__ slabs, int ldMax) {
    extern __shared__ float lds[];
    __shared__ int sTerm;
    __shared__ float sT_ds[24];
    __shared__ double sErr[NE_BLOCK / 64];
    const int c = blockIdx.y;
    if (threadIdx.x == 0) {
        int t = 0;
        while (t + 1 < P->nTerms && (int)blockIdx.x >= P->term[t + 1].blockStart) ++t;
        sTerm = t;
    }
    __syncthreads();
    const NoObsTerm &T = P->term[sTerm];
    const int nModels = P->nModels;
    float *poseC = poses + (size_t)c * P->poseStride;
    // stage every model's frame transforms: model mi at lds + sOff[mi] as [T_fm | T_mf]
    __shared__ int sOff[DARTK_MAX_MODELS + 1];
    if (threadIdx.x == 0) {
        int o = 0;
        for (int mi = 0; mi < nModels; ++mi) { sOff[mi] = o; o += 24 * P->model[mi].nFrames; }
        sOff[nModels] = o;
    }
    __syncthreads();
    for (int mi = 0; mi < nModels; ++mi) {
        float *p = lds + sOff[mi];
        const ModelView mv = rebase(P->model[mi], poseC + P->poseOff[mi]);
        const int n = 12 * mv.nFrames;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { p[i] = mv.T_fm[i]; p[n + i] = mv.T_mf[i]; }
    }
#define sTf(mi) (lds + sOff[mi])
    float *rows = lds + sOff[nModels];
    int *waveCnt = reinterpret_cast<int *>(rows + NE_BLOCK * ldMax);
    const int d = T.dims;
    const bool weighted = (T.type == TERM_PRIORS);
    const int ld = row_len(d, weighted);
    const ModelView src = rebase(P->model[T.src], poseC + P->poseOff[T.src]);
    const ModelView dst = rebase(P->model[T.dst], poseC + P->poseOff[T.dst]);
    if (T.type == TERM_INTER && threadIdx.x == 0) {      // T_ds = T_mc(dst) * T_cm(src), T_sd = inverse (optimizer.cpp:396-397)
        float A[12], B[12], T_ds[12], T_sd[12];
        for (int i = 0; i < 12; ++i) { A[i] = dst.T_mc[i]; B[i] = src.T_cm[i]; }
        se3_mul(A, B, T_ds);
        se3_invert(T_ds, T_sd);
        for (int i = 0; i < 12; ++i) { sT_ds[i] = T_ds[i]; sT_ds[12 + i] = T_sd[i]; }
    }
    __syncthreads();

    PairAcc<NE_BLOCK, PPT> acc;
    acc.init(d, false, weighted);
    double errInactive = 0.0;       // contact priors: _error of priors that contribute no row
    const int localBlock = blockIdx.x - T.blockStart;
    const int nTiles = (T.nPoints + NE_BLOCK - 1) / NE_BLOCK;
    for (int tile = localBlock; tile < nTiles; tile += T.nBlocks) {
        const int index = tile * NE_BLOCK + threadIdx.x;
        bool active = false;
        // values carried from the test phase to the row phase
        f3 pm; pm.x = pm.y = pm.z = 0.f;       // point in the source model frame
        f3 gs; gs.x = gs.y = gs.z = 0.f;       // gradient in the source model frame
        f3 pdm = pm, gdm = gs;                 // contact priors: point / gradient in the destination model frame
        int srcFrame = 0, dstFrame = 0, pSrc = 0, pDst = 0;
        float residual = 0.0f, weight = 0.0f, E = 0.0f;
        bool upSrc = true, upDst = false;
        if (index < T.nPoints) {
            if (T.type == TERM_SELF) {
                const float4 v = T.cloud[index];
                const int srcGrid = (int)roundf(v.w);
                srcFrame = src.sdfFrame[srcGrid];
                pm = se3_apply_w(sTf(T.src) + 12 * src.nFrames + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
                const Collision col = first_collision(src, sTf(T.src), pm, T.pot, srcGrid);
                if (col.hit) {
                    const SdfDev &s = src.sdfs[col.dstGrid];
                    const f3 gf = grid_gradient(src.grids + s.dataOff, s.dimx, s.dimy, col.vg);
                    gs = se3_rotate(sTf(T.src) + 12 * src.nFrames + 12 * src.sdfFrame[col.dstGrid], gf);
                    residual = col.residual;
                    active = true;
                }
            } else if (T.type == TERM_INTER) {
                const float4 v = T.cloud[index];
                const int srcGrid = (int)roundf(v.w);
                srcFrame = src.sdfFrame[srcGrid];
                pm = se3_apply_w(sTf(T.src) + 12 * src.nFrames + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
                const f3 vdm = se3_apply_w(sT_ds, pm.x, pm.y, pm.z, 1.0f);
                const Collision col = first_collision(dst, sTf(T.dst), vdm, nullptr, 0);
                if (col.hit) {
                    const SdfDev &s = dst.sdfs[col.dstGrid];
                    const f3 gf = grid_gradient(dst.grids + s.dataOff, s.dimx, s.dimy, col.vg);
                    const f3 gd = se3_rotate(sTf(T.dst) + 12 * dst.nFrames + 12 * dst.sdfFrame[col.dstGrid], gf);
                    gs = se3_rotate(sT_ds + 12, gd);
                    residual = col.residual;
                    active = true;
                }
            } else {
                // ContactPrior::computeContribution, contact_prior.cpp:17-95
                const dartk_contact_prior pr = P->priors[index];
                const int sign = pr.invert ? -1 : 1;
                E = sign * pr.weight * pr.contactThreshold;
                if (pr.weight != 0) {
                    pSrc = pr.srcModel; pDst = pr.dstModel;
                    const ModelView ms = rebase(P->model[pSrc], poseC + P->poseOff[pSrc]);
                    const ModelView md = rebase(P->model[pDst], poseC + P->poseOff[pDst]);
                    const float *sfm = sTf(pSrc), *smf = sTf(pSrc) + 12 * ms.nFrames;
                    const float *dfm = sTf(pDst), *dmf = sTf(pDst) + 12 * md.nFrames;
                    srcFrame = ms.sdfFrame[pr.srcSdf];
                    pm = se3_apply_w(smf + 12 * srcFrame, pr.contactPoint[0], pr.contactPoint[1], pr.contactPoint[2], 1.0f);
                    const f3 cc = se3_apply_w(ms.T_cm, pm.x, pm.y, pm.z, 1.0f);
                    pdm = se3_apply_w(md.T_mc, cc.x, cc.y, cc.z, 1.0f);
                    // nearest destination SDF by raw voxel-unit value among isInBoundsInterp grids (:41-54; the
                    // reference loop bound `<= getNumSdfs()` reads one element past the end -- restated with `<`)
                    int dstSdf = -1;
                    float minV = 3.402823466e+38f;
                    for (int ds = 0; ds < md.nSdfs; ++ds) {
                        const f3 df = se3_apply_w(dfm + 12 * md.sdfFrame[ds], pdm.x, pdm.y, pdm.z, 1.0f);
                        const SdfDev &s = md.sdfs[ds];
                        const f3 g = grid_coords(s, df);
                        if (!in_bounds_interp(s, g)) continue;
                        const float v = grid_value(md.grids + s.dataOff, s.dimx, s.dimy, g);
                        if (v < minV) { minV = v; dstSdf = ds; }
                    }
                    if (dstSdf >= 0) {
                        dstFrame = md.sdfFrame[dstSdf];
                        const SdfDev &s = md.sdfs[dstSdf];
                        const f3 cdf = se3_apply_w(dfm + 12 * dstFrame, pdm.x, pdm.y, pdm.z, 1.0f);
                        const f3 cg = grid_coords(s, cdf);
                        if (in_bounds_grad_interp(s, cg)) {
                            float err = grid_value(md.grids + s.dataOff, s.dimx, s.dimy, cg) * s.res;
                            if (!(fabsf(err) > pr.contactThreshold)) {
                                if ((double)fabsf(err) < 1e-4) E = 0.0f;       // :70-73
                                else {
                                    const f3 g_df = grid_gradient(md.grids + s.dataOff, s.dimx, s.dimy, cg);
                                    gdm = se3_rotate(dmf + 12 * dstFrame, g_df);
                                    const f3 g_c = se3_rotate(md.T_cm, gdm);
                                    gs = se3_rotate(ms.T_mc, g_c);
                                    bool keep = true;
                                    if (pr.invert && err > 0) {            // repulsive cylinder test, :84-89
                                        const f3 g_sf = se3_rotate(sfm + 12 * srcFrame, gs);
                                        const float invLen = 1.0f / sqrtf(dot(g_sf, g_sf));
                                        f3 n; n.x = g_sf.x * invLen; n.y = g_sf.y * invLen; n.z = g_sf.z * invLen;
                                        f3 nsf; nsf.x = pr.normal[0]; nsf.y = pr.normal[1]; nsf.z = pr.normal[2];
                                        const float cosT = dot(nsf, n);
                                        const float sinT = sqrtf(1.f - cosT * cosT);
                                        if (fabsf(err * sinT) > 0.01f) keep = false;
                                    }
                                    if (keep) {
                                        if (pr.invert) err *= -1.f;
                                        E = sign * pr.weight * fabsf(err);
                                        residual = err; weight = pr.weight;
                                        upSrc = pr.updateSrc != 0; upDst = pr.updateDst != 0;
                                        active = upSrc || upDst;
                                    }
                                }
                            }
                        }
                    }
                }
                if (!active) errInactive += (double)E;
            }
        }
        int nActive;
        const int slot = compact_slot<NE_BLOCK>(active, waveCnt, nActive);
        if (active) {
            float *row = rows + slot * ld;
            if (T.type == TERM_SELF) {
                error_jacobian<true>(row, pm, 1.0f, src.depMask[srcFrame], gs, src.nJoints, src.jointType, src.jointAxis,
                                     sTf(T.src), sTf(T.src) + 12 * src.nFrames);
                row[d] = residual;
                row[d + 1] = (float)(0.5 * (double)residual * (double)residual);
                row[d + 2] = 1.0f;
            } else if (T.type == TERM_INTER) {
                error_jacobian<false>(row, pm, 1.0f, src.depMask[srcFrame], gs, src.nJoints, src.jointType, src.jointAxis,
                                      sTf(T.src), sTf(T.src) + 12 * src.nFrames);
                row[d] = residual;
                row[d + 1] = (float)(0.5 * (double)residual * (double)residual);
                row[d + 2] = 1.0f;
            } else {
                // subJ = [ dot(grad_sm, J3D_src) | -dot(grad_dm, J3D_dst) ] placed at the models' global offsets (:110-126)
                for (int i = 0; i < d; ++i) row[i] = 0.0f;
                const ModelView &ms = P->model[pSrc];
                const ModelView &md = P->model[pDst];
                if (upSrc)
                    error_jacobian<false>(row + P->off[pSrc], pm, 1.0f, ms.depMask[srcFrame], gs, ms.nJoints, ms.jointType, ms.jointAxis,
                                          sTf(pSrc), sTf(pSrc) + 12 * ms.nFrames);
                if (upDst) {
                    float *rd = row + P->off[pDst];
                    error_jacobian<false>(rd, pdm, 1.0f, md.depMask[dstFrame], gdm, md.nJoints, md.jointType, md.jointAxis,
                                          sTf(pDst), sTf(pDst) + 12 * md.nFrames);
                    const int nd = 6 + md.nJoints;
                    for (int i = 0; i < nd; ++i) rd[i] = -rd[i];
                }
                row[d] = weight * residual;         // J^T e += (w*err) * J   (:128)
                row[d + 1] = E;
                row[d + 2] = 1.0f;
                for (int i = 0; i < d; ++i) row[d + 3 + i] = weight * row[i];      // (w*J) J^T   (:129)
            }
        }
        __syncthreads();
        if (nActive) acc.accumulate(rows, ld, nActive);
        __syncthreads();
    }
    if (T.type == TERM_PRIORS) {    // add the _error of the row-less priors to the e slot, fixed order
        double v = errInactive;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sErr[threadIdx.x >> 6] = v;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NE_BLOCK / 64; ++w) tot += sErr[w];
        acc.add_at(T.S - 1, tot);
    }
    float *slab = slabs + (size_t)c * P->slabStride + T.slabOff + (size_t)localBlock * T.S;
    acc.store(slab, T.S);
#undef sTf
}

// -----------------------------------------------------------------------------------------------------
// per candidate: reduce + assemble + solve + update + FK (optimizer.cpp:853-954), or the energy only (getError)
// -----------------------------------------------------------------------------------------------------
struct NoObsSolveArgs {
    float regularization[DARTK_MAX_MODELS], regularizationScaled[DARTK_MAX_MODELS];
    float lambda[DARTK_MAX_MODELS * DARTK_MAX_MODELS];
    int energyOnly;
    int iteration, numIterations;
    float *energy;        // [batch]
    float *iterInfo;      // [batch][numIterations][8] or null
    float *firstSystem;   // [batch][D*D + D] or null (iteration 0 only)
};

__device__ __forceinline__ float reduce_slot(const float *slabs, int nBlocks, int S, int o) {
    double s = 0.0;
    for (int b = 0; b < nBlocks; ++b) s += (double)slabs[(size_t)b * S + o];
    return (float)s;
}

__global__ void __launch_bounds__(64)
noobs_solve_kernel(const NoObsProblem *__restrict__ P, float *__restrict__ poses, const float *__restrict__ slabs, NoObsSolveArgs a) {
    extern __shared__ float lds[];
    const int c = blockIdx.x;
    const int D = P->D, nModels = P->nModels;
    const int ld = (D + 1) | 1;
    float *U = lds;                  // [D][ld] upper triangle of the global system
    float *b = U + D * ld;           // [D]
    float *sys = b + D;              // reduced packed system of the current term
    const int lane = threadIdx.x;
    const float *slabC = slabs + (size_t)c * P->slabStride;
    float *poseC = poses + (size_t)c * P->poseStride;
    for (int i = lane; i < D * ld; i += 64) U[i] = 0.0f;
    for (int i = lane; i < D; i += 64) b[i] = 0.0f;
    float eInt = 0.0f, ePri = 0.0f;
    for (int t = 0; t < P->nTerms; ++t) {
        const NoObsTerm &T = P->term[t];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int o = lane; o < T.S; o += 64) sys[o] = reduce_slot(slabC + T.slabOff, T.nBlocks, T.S, o);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int d = T.dims;
        if (T.type == TERM_PRIORS) {
            // priors add into the global upper triangle after the model blocks were scattered (:913-916)
            for (int idx = lane; idx < D * D; idx += 64) {
                const int i = idx / D, j = idx - i * D;
                if (j < i) continue;
                const float v = sys[d + ((j * (j + 1)) >> 1) + i];
                if (v != 0) U[i * ld + j] += v;
            }
            for (int i = lane; i < D; i += 64) b[i] += sys[i];
            ePri = sys[T.S - 1];
        } else {
            const int m = T.src;
            const float lam = a.lambda[T.src + T.dst * nModels];
            const int base = P->off[m] + (T.type == TERM_SELF ? 6 : 0);     // bottom-right block for the articulated system
            // unpack (optimizer.cpp:50-70): eJ += lam*J^T e, upper JTJ += lam*packed, e = lam*e
            for (int idx = lane; idx < d * d; idx += 64) {
                const int i = idx / d, j = idx - i * d;
                if (j < i) continue;
                U[(base + i) * ld + base + j] += lam * sys[d + ((j * (j + 1)) >> 1) + i];
            }
            for (int i = lane; i < d; i += 64) b[base + i] += lam * sys[i];
            eInt += lam * sys[T.S - 1];       // getError sums the terms (:1047,1055)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float energy = eInt + (P->nPriors > 0 ? ePri / (float)P->nPriors : 0.0f);
    if (a.energyOnly) {
        if (lane == 0) a.energy[c] = energy;
        return;
    }
    if (a.iterInfo && lane == 0) {
        float *r = a.iterInfo + ((size_t)c * a.numIterations + a.iteration) * 8;
        r[0] = eInt; r[1] = ePri; r[2] = energy; r[3] = r[4] = r[5] = r[6] = r[7] = 0.0f;
    }
    if (a.firstSystem && a.iteration == 0) {
        float *fs = a.firstSystem + (size_t)c * (D * D + D);
        for (int idx = lane; idx < D * D; idx += 64) fs[idx] = U[(idx / D) * ld + (idx % D)];
        for (int i = lane; i < D; i += 64) fs[D * D + i] = b[i];
    }
    // regularisation and LM damping of the model blocks, after the priors (:919-929)
    for (int mi = 0; mi < nModels; ++mi) {
        const int off = P->off[mi];
        for (int i = lane; i < P->dims[mi]; i += 64) {
            float v = U[(off + i) * ld + off + i];
            v += a.regularization[mi];
            v += a.regularizationScaled[mi] * v;
            U[(off + i) * ld + off + i] = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // paramUpdate = -U^-1 JTe by back substitution on the upper triangle (Eigen triangularView<Upper>().solve on a
    // column-major sparse matrix: columns last to first, a column is skipped when its rhs entry is exactly 0)
    float x = lane < D ? b[lane] : 0.0f;      // D <= 64: lane r holds x[r]
    for (int i = D - 1; i >= 0; --i) {
        float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
        if (xi != 0.0f) {
            xi = xi / U[i * ld + i];
            if (lane == i) x = xi;
            if (lane < i) {
                const float u = U[lane * ld + i];
                if (u != 0.0f) x -= xi * u;
            }
        }
    }
    x = -x;
    if (lane < D) b[lane] = x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pose update (:933-954) and forward kinematics for the next iteration
    float *sT = sys;     // FK scratch (12 * nFrames floats), the packed system is no longer needed
    for (int mi = 0; mi < nModels; ++mi) {
        const ModelView mv = rebase(P->model[mi], poseC + P->poseOff[mi]);
        const int off = P->off[mi];
        if (lane == 0) {
            float t6[6], dT[12], T[12], Tn[12], Tinv[12];
            for (int i = 0; i < 6; ++i) t6[i] = b[off + i];
            se3_exp(t6, dT);
            for (int i = 0; i < 12; ++i) T[i] = mv.T_mc[i];
            se3_mul(dT, T, Tn);
            se3_invert(Tn, Tinv);
            for (int i = 0; i < 12; ++i) { mv.T_mc[i] = Tn[i]; mv.T_cm[i] = Tinv[i]; }
        }
        if (lane < mv.nJoints) {
            const float v = mv.joints[lane] + b[off + 6 + lane];
            mv.joints[lane] = fminf(fmaxf(mv.limits[2 * lane], v), mv.limits[2 * lane + 1]);
        }
        fk_wave(mv, 0, sT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// export poses of the batch
__global__ void noobs_store_poses_kernel(const NoObsProblem *__restrict__ P, const float *__restrict__ poses, float *__restrict__ T_mcs,
                                         float *__restrict__ joints) {
    const int mi = blockIdx.x, c = blockIdx.y;
    const float *pose = poses + (size_t)c * P->poseStride + P->poseOff[mi];
    const int lane = threadIdx.x;
    if (lane < 12) T_mcs[((size_t)c * P->nModels + mi) * 12 + lane] = pose[lane];
    if (lane < P->model[mi].nJoints) joints[(size_t)c * P->jointStride + P->jointOff[mi] + lane] = pose[24 + lane];
}

struct NoObs {
    NoObsProblem h;            // host copy
    NoObsProblem *d;           // device copy
    dartk_contact_prior *d_priors;
    int nPriors;
    int maxBatch;
    float *poses, *slabs;
    int totalBlocks, ldMax, SMax, maxFrames;
    size_t ldsTerms, ldsSolve;
    bool lambdaActive[DARTK_MAX_MODELS * DARTK_MAX_MODELS];
};

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_noobs_destroy(dartk_noobs_t h) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    if (!t) return DARTK_OK;
    if (t->d) (void)hipFree(t->d);
    if (t->d_priors) (void)hipFree(t->d_priors);
    if (t->poses) (void)hipFree(t->poses);
    if (t->slabs) (void)hipFree(t->slabs);
    delete t;
    return DARTK_OK;
}

int dartk_noobs_create(int nModels, const dartk_model_t *models, const float *const *d_clouds, const int32_t *nCloudPoints,
                       const int32_t *const *d_potentials, const float *h_lambdaIntersection, const dartk_contact_prior *h_priors,
                       int nPriors, int maxBatch, dartk_noobs_t *out) {
    DARTK_REQUIRE(models && out && h_lambdaIntersection && nCloudPoints && d_clouds && d_potentials, "dartk_noobs_create: null argument");
    DARTK_REQUIRE(nModels >= 1 && nModels <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", nModels, DARTK_MAX_MODELS);
    DARTK_REQUIRE(nPriors >= 0 && (nPriors == 0 || h_priors), "bad prior list");
    DARTK_REQUIRE(maxBatch >= 1, "maxBatch must be >= 1");
    NoObs *t = new NoObs();
    memset(t, 0, sizeof(NoObs));
    NoObsProblem &P = t->h;
    P.nModels = nModels; P.nPriors = nPriors;
    int D = 0, poseOff = 0, jointOff = 0, maxFrames = 1;
    for (int m = 0; m < nModels; ++m) {
        Model *mod = reinterpret_cast<Model *>(models[m]);
        if (!mod) { delete t; set_error("dartk_noobs_create: null model %d", m); return DARTK_ERR_INVALID; }
        P.model[m] = mod->v;
        P.off[m] = D; P.dims[m] = 6 + mod->v.nJoints; D += P.dims[m];
        P.poseOff[m] = poseOff; poseOff += pose_block_floats(mod->v);
        P.jointOff[m] = jointOff; jointOff += mod->v.nJoints;
        if (mod->v.nFrames > maxFrames) maxFrames = mod->v.nFrames;
    }
    if (D > 64) { delete t; set_error("global system dimension %d exceeds 64 (one lane per unknown in the solve)", D); return DARTK_ERR_UNSUPPORTED; }
    P.D = D; P.poseStride = poseOff; P.jointStride = jointOff > 0 ? jointOff : 1;
    for (int p = 0; p < nPriors; ++p) {
        const dartk_contact_prior &q = h_priors[p];
        if (q.srcModel < 0 || q.srcModel >= nModels || q.dstModel < 0 || q.dstModel >= nModels || q.srcSdf < 0 ||
            q.srcSdf >= P.model[q.srcModel].nSdfs) {
            delete t; set_error("contact prior %d: model / SDF index out of range", p); return DARTK_ERR_INVALID;
        }
        if (q.srcModel == q.dstModel) { delete t; set_error("contact prior %d: source and destination model must differ", p); return DARTK_ERR_UNSUPPORTED; }
        if (q.updateContact) {
            delete t;
            set_error("contact prior %d: update_contact_point is not supported on the device path (ContactGrasp freezes the contact points, "
                      "grasp_analyzer.cpp:359,374,383)", p);
            return DARTK_ERR_UNSUPPORTED;
        }
    }
    // term table in the reference's evaluation order: per model self-intersection, then the other models; priors last
    const bool small = maxBatch < 64;     // few candidates: one workgroup per tile; many: a few workgroups per term
    int nTerms = 0, block = 0, slabOff = 0, SMax = 0, ldMax = 0;
    auto add = [&](int type, int src, int dst, int nPoints, int dims, const float *cloud, const int *pot) {
        NoObsTerm &T = P.term[nTerms++];
        T.type = type; T.src = src; T.dst = dst; T.nPoints = nPoints; T.dims = dims; T.S = sys_size(dims);
        const int tiles = (nPoints + NE_BLOCK - 1) / NE_BLOCK;
        T.nBlocks = small ? (tiles > 0 ? tiles : 1) : (tiles < 4 ? (tiles > 0 ? tiles : 1) : 4);
        T.blockStart = block; block += T.nBlocks;
        T.slabOff = slabOff; slabOff += T.nBlocks * T.S;
        T.cloud = reinterpret_cast<const float4 *>(cloud); T.pot = pot;
        if (T.S > SMax) SMax = T.S;
        const int ld = row_len(dims, type == TERM_PRIORS);
        if (ld > ldMax) ldMax = ld;
    };
    for (int m = 0; m < nModels; ++m) {
        if (h_lambdaIntersection[m + m * nModels] > 0 && nCloudPoints[m] > 0 && P.dims[m] > 6) {
            if (!d_clouds[m] || !d_potentials[m]) { delete t; set_error("model %d: self-intersection needs a cloud and a potential matrix", m); return DARTK_ERR_INVALID; }
            add(TERM_SELF, m, m, nCloudPoints[m], P.dims[m] - 6, d_clouds[m], d_potentials[m]);
        }
        for (int d = 0; d < nModels; ++d) {
            if (d == m || !(h_lambdaIntersection[m + d * nModels] > 0) || nCloudPoints[m] <= 0) continue;
            if (!d_clouds[m]) { delete t; set_error("model %d: intersection needs a collision cloud", m); return DARTK_ERR_INVALID; }
            add(TERM_INTER, m, d, nCloudPoints[m], P.dims[m], d_clouds[m], nullptr);
        }
    }
    if (nPriors > 0) add(TERM_PRIORS, 0, 0, nPriors, D, nullptr, nullptr);
    for (int i = 0; i < nModels * nModels; ++i) t->lambdaActive[i] = h_lambdaIntersection[i] > 0;
    P.nTerms = nTerms; P.slabStride = slabOff > 0 ? slabOff : 1;
    t->totalBlocks = block; t->SMax = SMax; t->ldMax = ldMax; t->maxFrames = maxFrames; t->maxBatch = maxBatch; t->nPriors = nPriors;
    int ldsModels = 0;
    for (int m = 0; m < nModels; ++m) ldsModels += 24 * P.model[m].nFrames;
    t->ldsTerms = sizeof(float) * ((size_t)ldsModels + (size_t)NE_BLOCK * ldMax + 8);
    const int ldU = (D + 1) | 1;
    const int sysFloats = SMax > 12 * maxFrames ? SMax : 12 * maxFrames;
    t->ldsSolve = sizeof(float) * ((size_t)D * ldU + D + sysFloats + 8);
    if (t->ldsTerms > 160 * 1024 || t->ldsSolve > 160 * 1024) { delete t; set_error("LDS requirement exceeds 160 KiB"); return DARTK_ERR_UNSUPPORTED; }
    hipError_t e = hipSuccess;
    if (nPriors > 0) {
        e = hipMalloc((void **)&t->d_priors, sizeof(dartk_contact_prior) * (size_t)nPriors);
        if (e == hipSuccess) e = hipMemcpy(t->d_priors, h_priors, sizeof(dartk_contact_prior) * (size_t)nPriors, hipMemcpyHostToDevice);
    }
    P.priors = t->d_priors;
    if (e == hipSuccess) e = hipMalloc((void **)&t->d, sizeof(NoObsProblem));
    if (e == hipSuccess) e = hipMemcpy(t->d, &P, sizeof(NoObsProblem), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&t->poses, sizeof(float) * (size_t)P.poseStride * maxBatch);
    if (e == hipSuccess) e = hipMalloc((void **)&t->slabs, sizeof(float) * (size_t)P.slabStride * maxBatch);
    if (e != hipSuccess) {
        set_error("dartk_noobs_create: %s", hipGetErrorString(e));
        dartk_noobs_destroy(reinterpret_cast<dartk_noobs_t>(t));
        return DARTK_ERR_HIP;
    }
    *out = reinterpret_cast<dartk_noobs_t>(t);
    return DARTK_OK;
}

int dartk_noobs_set_priors(dartk_noobs_t h, const dartk_contact_prior *h_priors, int nPriors, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && (h_priors || nPriors == 0), "dartk_noobs_set_priors: null argument");
    DARTK_REQUIRE(nPriors == t->nPriors, "the number of priors (%d) is fixed at creation (%d)", nPriors, t->nPriors);
    for (int p = 0; p < nPriors; ++p)
        DARTK_REQUIRE(!h_priors[p].updateContact, "contact prior %d: update_contact_point is not supported on the device path", p);
    if (nPriors) {
        DARTK_HIP(hipMemcpyAsync(t->d_priors, h_priors, sizeof(dartk_contact_prior) * (size_t)nPriors, hipMemcpyHostToDevice, as_stream(stream)));
        DARTK_HIP(hipStreamSynchronize(as_stream(stream)));
    }
    return DARTK_OK;
}

static int noobs_launch_terms(NoObs *t, int batch, hipStream_t st) {
    const dim3 grid(t->totalBlocks, batch), block(NE_BLOCK);
    if (t->totalBlocks == 0) return DARTK_OK;
    ProfScope ps(DARTK_PROF_NOOBS_TERMS, st);
#define DARTK_LAUNCH_TERMS(PPT_)                                                                                 \
    do {                                                                                                         \
        auto kfn = noobs_terms_kernel<PPT_>;                                                                     \
        if (t->ldsTerms > 64 * 1024)                                                                             \
            DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->ldsTerms)); \
        hipLaunchKernelGGL(kfn, grid, block, t->ldsTerms, st, t->d, t->poses, t->slabs, t->ldMax);               \
    } while (0)
    if (t->SMax <= 3 * NE_BLOCK) DARTK_LAUNCH_TERMS(3);
    else if (t->SMax <= 6 * NE_BLOCK) DARTK_LAUNCH_TERMS(6);
    else DARTK_LAUNCH_TERMS(20);
#undef DARTK_LAUNCH_TERMS
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

static int noobs_check_opts(NoObs *t, const dartk_noobs_options *o) {
    DARTK_REQUIRE(o->nModels == t->h.nModels, "options are for %d models, the tracker has %d", o->nModels, t->h.nModels);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    for (int i = 0; i < o->nModels * o->nModels; ++i)
        DARTK_REQUIRE((o->lambdaIntersection[i] > 0) == t->lambdaActive[i],
                      "lambdaIntersection[%d] switches a term on/off relative to dartk_noobs_create (terms are fixed at creation)", i);
    return DARTK_OK;
}

static void noobs_fill_args(NoObs *t, const dartk_noobs_options *o, NoObsSolveArgs &a) {
    memset(&a, 0, sizeof(a));
    for (int m = 0; m < o->nModels; ++m) { a.regularization[m] = o->regularization[m]; a.regularizationScaled[m] = o->regularizationScaled[m]; }
    for (int i = 0; i < o->nModels * o->nModels; ++i) a.lambda[i] = o->lambdaIntersection[i];
    a.numIterations = o->numIterations;
}

int dartk_noobs_optimize_poses(dartk_noobs_t h, const dartk_noobs_options *o, int batch, float *d_T_mcs, float *d_joints,
                               float *d_energy, float *d_iterInfo, float *d_firstSystem, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && o && d_T_mcs && d_energy, "dartk_noobs_optimize_poses: null argument");
    DARTK_REQUIRE(batch >= 1 && batch <= t->maxBatch, "batch %d out of range [1,%d]", batch, t->maxBatch);
    DARTK_REQUIRE(d_joints || t->h.jointStride <= 1, "joint array required");
    int rc = noobs_check_opts(t, o);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (t->ldsSolve > 64 * 1024)
        DARTK_HIP(hipFuncSetAttribute((const void *)noobs_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->ldsSolve));
    hipLaunchKernelGGL(noobs_load_poses_kernel, dim3(t->h.nModels, batch), dim3(64), 0, st, t->d, t->poses, d_T_mcs, d_joints);
    NoObsSolveArgs a;
    noobs_fill_args(t, o, a);
    a.energy = d_energy; a.iterInfo = d_iterInfo; a.firstSystem = d_firstSystem;
    for (int it = 0; it < o->numIterations; ++it) {
        rc = noobs_launch_terms(t, batch, st);
        if (rc) return rc;
        a.iteration = it; a.energyOnly = 0;
        ProfScope ps(DARTK_PROF_NOOBS_SOLVE, st);
        hipLaunchKernelGGL(noobs_solve_kernel, dim3(batch), dim3(64), t->ldsSolve, st, t->d, t->poses, t->slabs, a);
    }
    // getError at the final poses (optimizer.cpp:980-982)
    rc = noobs_launch_terms(t, batch, st);
    if (rc) return rc;
    a.energyOnly = 1;
    hipLaunchKernelGGL(noobs_solve_kernel, dim3(batch), dim3(64), t->ldsSolve, st, t->d, t->poses, t->slabs, a);
    hipLaunchKernelGGL(noobs_store_poses_kernel, dim3(t->h.nModels, batch), dim3(64), 0, st, t->d, t->poses, d_T_mcs, d_joints);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_noobs_get_error(dartk_noobs_t h, const dartk_noobs_options *o, int batch, const float *d_T_mcs, const float *d_joints,
                          float *d_energy, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && o && d_T_mcs && d_energy, "dartk_noobs_get_error: null argument");
    DARTK_REQUIRE(batch >= 1 && batch <= t->maxBatch, "batch %d out of range [1,%d]", batch, t->maxBatch);
    int rc = noobs_check_opts(t, o);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (t->ldsSolve > 64 * 1024)
        DARTK_HIP(hipFuncSetAttribute((const void *)noobs_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->ldsSolve));
    hipLaunchKernelGGL(noobs_load_poses_kernel, dim3(t->h.nModels, batch), dim3(64), 0, st, t->d, t->poses, d_T_mcs, d_joints);
    rc = noobs_launch_terms(t, batch, st);
    if (rc) return rc;
    NoObsSolveArgs a;
    noobs_fill_args(t, o, a);
    a.energy = d_energy; a.energyOnly = 1;
    hipLaunchKernelGGL(noobs_solve_kernel, dim3(batch), dim3(64), t->ldsSolve, st, t->d, t->poses, t->slabs, a);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

}  // extern "C"
