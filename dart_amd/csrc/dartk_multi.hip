// dartk_multi.hip -- Optimizer::optimizePoses (optimizer.cpp:524-781), device-resident: multi-model tracking with
// observations and (optionally) reduced poses. The reference drives every term from the host with one blocking
// device->host copy per term, assembles and solves in Eigen, and uploads every pose again; here one call enqueues all
// iterations: the terms leave their packed systems in HBM, one wave assembles the block system in the reference's order,
// applies regularisation + LM scaling, runs the reference's upper-triangular back substitution (SURVEY A-1), updates the
// reduced poses with their limits, projects them to full joint vectors and runs forward kinematics. No host round trip.
#include "dartk_internal.h"

#include <cstdlib>
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_fk.h"
#include "dartk_raycast.h"

#include <string.h>
#include <vector>

namespace dartk {

struct MultiSolveModel {
    ModelView v;
    int redType, redArt, dims, off;         // dims = 6 + redArt; off = offset in the global system
    const float *A;                         // linear reduction: [nJoints][redArt]
    const int *map;                         // parameter map: [nJoints]
    const float *b, *redMin, *redMax;       // device copies
    float *redArtPtr;                       // [redArt] reduced articulation (device, in/out)
    const float *sysA, *sysB, *sysC;        // packed systems of this iteration (null = term off)
    const float *sysI[DARTK_MAX_MODELS];
    float lamSelf, lamI[DARTK_MAX_MODELS];
    const int *countA, *countB;
    float reg, regScaled;
    const float *damping;                   // [dims][dims] or null (Tracker::getDampingMatrix)
};
struct MultiSolveArgs {
    int nModels, D, numIterations;
    float lambdaA, lambdaB;
    float *iterInfo;        // [numIterations][nModels][8] or null
    float *firstSystem;     // [D*D + D] or null
    MultiSolveModel m[DARTK_MAX_MODELS];
    // batches of independent frames (workgroup = frame): what lies between two frames in the tables the pointers above
    // name (systems, counts: FrameBatch; reduced articulations; iteration summaries). All 0 for the single-frame loop.
    FrameBatch fb;
    int redStride;
    size_t infoStride;
    // batch only: the caller's pose arrays, read before the first iteration / written after the last (iteration == -1 / -2)
    float *T_mcs;           // [frames][nModels][12]
    // contact priors of the single-frame loop (optimizer.cpp:727-732, 757-759): device array, contact points updated in place
    dartk_contact_prior *priors;
    int nPriors;
    // the association / prediction counters of a frame (fb.counts ints each): cleared here for the next iteration's kernels
    // instead of by two memset launches per iteration (null: the term is off)
    int *zeroA, *zeroB;
    int *itCounter;         // every solve leaves its iteration number + 1 here (read by a launch with iteration == -3)
};

// the rows of the reduction a lane needs for its joint (pose.cpp:66-73, 91-97), loaded ahead of the solve like the FK tables
static constexpr int kRedPrefetch = 8;
struct RedTables { float b, A[kRedPrefetch]; int map; };
__device__ __forceinline__ RedTables red_prefetch(const MultiSolveModel &m, const ModelView &v) {
    const int lane = threadIdx.x & 63;
    RedTables t;
    t.b = 0.0f; t.map = 0;
#pragma unroll
    for (int r = 0; r < kRedPrefetch; ++r) t.A[r] = 0.0f;
    if (lane < v.nJoints) {
        if (m.redType == 2) t.map = m.map[lane];
        else if (m.redType == 1) {
            t.b = m.b[lane];
#pragma unroll
            for (int r = 0; r < kRedPrefetch; ++r) if (r < m.redArt) t.A[r] = m.A[r + lane * m.redArt];
        }
    }
    return t;
}
// projection of the reduced articulation in ra[] (LDS) to the full joint vector (pose.cpp:66-73, 91-97) and forward
// kinematics (Tracker::updatePose) of one model; one wave. `t`: the model's constant joint / frame tables (fk_prefetch), loaded by
// the caller ahead of the work the new joint values wait for. Joint j belongs to lane j, which is the lane that moves frame j + 1:
// its value goes into the kinematics in a register, the store to v.joints is for the kernels that follow.
__device__ __forceinline__ void project_and_fk(const MultiSolveModel &m, const ModelView &v, const float *ra, float *sT, const FkTables &t,
                                               const RedTables &rt) {
    const int lane = threadIdx.x & 63;
    float full = 0.0f;
    if (lane < v.nJoints) {
        if (m.redType == 0) full = ra[lane];
        else if (m.redType == 2) full = ra[rt.map];
        else {
            full = rt.b;
#pragma unroll
            for (int r = 0; r < kRedPrefetch; ++r) if (r < m.redArt) full = full + ra[r] * rt.A[r];                 // pose.cpp:66-73
            for (int r = kRedPrefetch; r < m.redArt; ++r) full = full + ra[r] * m.A[r + lane * m.redArt];
        }
        v.joints[lane] = full;
    }
    fk_wave(v, 0, sT, t, (lane + 1 < v.nFrames) ? full : 0.0f);
}

__device__ __forceinline__ void wave_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ContactPrior::computeContribution (contact_prior.cpp:8-205) and updatePriorParams (:207-221) of one prior, by ONE lane: the
// tracking loop carries a handful of priors (hand-object contacts), so they are evaluated in the solve kernel against the
// poses of this iteration instead of in a term kernel of their own (ContactGrasp's thousands go through dartk_noobs).
// Writes the prior's Jacobian row over the GLOBAL unknowns into L[0..D) and sc = {J^T e factor, weight, row is live};
// moves the prior's contact point when it has update_contact_point.
__device__ __noinline__ void multi_eval_contact_prior(const MultiSolveArgs &a, int frame, dartk_contact_prior *prp, float *L, float *sc) {
    const dartk_contact_prior pr = *prp;
    sc[0] = 0.0f; sc[1] = 0.0f; sc[2] = 0.0f;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;          // solution of the prior's own three unknowns (update_contact_point)
    for (int i = 0; i < a.D; ++i) L[i] = 0.0f;
    ModelView ms = a.m[pr.srcModel].v, md = a.m[pr.dstModel].v;
    select_frame(ms, frame); select_frame(md, frame);
    do {
        if (pr.weight == 0) break;
        const int srcFrame = ms.sdfFrame[pr.srcSdf];
        const f3 pm = se3_apply_w(ms.T_mf + 12 * srcFrame, pr.contactPoint[0], pr.contactPoint[1], pr.contactPoint[2], 1.0f);
        const f3 cc = se3_apply_w(ms.T_cm, pm.x, pm.y, pm.z, 1.0f);
        const f3 pdm = se3_apply_w(md.T_mc, cc.x, cc.y, cc.z, 1.0f);
        // nearest destination SDF by raw voxel-unit value among the isInBoundsInterp grids (:41-54; the reference's loop bound
        // `<= getNumSdfs()` reads one element past the end -- restated with `<`)
        int dstSdf = -1;
        float minV = 3.402823466e+38f;
        for (int ds = 0; ds < md.nSdfs; ++ds) {
            const SdfDev &s = md.sdfs[ds];
            const f3 df = se3_apply_w(md.T_fm + 12 * md.sdfFrame[ds], pdm.x, pdm.y, pdm.z, 1.0f);
            const f3 g = grid_coords(s, df);
            if (!in_bounds_interp(s, g)) continue;
            const float v = sdf_value<CNT_PRIOR>(md.grids, s, g);
            if (v < minV) { minV = v; dstSdf = ds; }
        }
        if (dstSdf < 0) break;
        const int dstFrame = md.sdfFrame[dstSdf];
        const SdfDev &s = md.sdfs[dstSdf];
        const f3 cg = grid_coords(s, se3_apply_w(md.T_fm + 12 * dstFrame, pdm.x, pdm.y, pdm.z, 1.0f));
        if (!in_bounds_grad_interp(s, cg)) break;
        float err = sdf_value<CNT_PRIOR>(md.grids, s, cg) * s.res;
        if (fabsf(err) > pr.contactThreshold) break;
        if ((double)fabsf(err) < 1e-4) break;                           // :70-73
        const f3 g_df = sdf_gradient<CNT_PRIOR>(md.grids, s, cg);
        const f3 gdm = se3_rotate(md.T_mf + 12 * dstFrame, g_df);
        const f3 g_c = se3_rotate(md.T_cm, gdm);
        const f3 gsm = se3_rotate(ms.T_mc, g_c);
        const f3 g_sf = se3_rotate(ms.T_fm + 12 * srcFrame, gsm);
        if (pr.invert && err > 0) {                                     // repulsive cylinder test, :84-89
            const float invLen = 1.0f / sqrtf(dot(g_sf, g_sf));
            f3 n; n.x = g_sf.x * invLen; n.y = g_sf.y * invLen; n.z = g_sf.z * invLen;
            f3 nsf; nsf.x = pr.normal[0]; nsf.y = pr.normal[1]; nsf.z = pr.normal[2];
            const float cosT = dot(nsf, n);
            const float sinT = sqrtf(1.f - cosT * cosT);
            if (fabsf(err * sinT) > 0.01f) break;
        }
        if (pr.invert) err *= -1.f;
        if (!(pr.updateContact || pr.updateSrc || pr.updateDst)) break;
        // the priors index the model Jacobian by the pose's REDUCED dimensions (:31-32,110-126): the first dims columns of the
        // full Jacobian, as the reference computes it
        const MultiSolveModel &sm = a.m[pr.srcModel], &dm = a.m[pr.dstModel];
        if (pr.updateSrc)
            error_jacobian<false>(L + sm.off, pm, 1.0f, ms.depMask[srcFrame], gsm, sm.dims - 6, ms.jointType, ms.jointAxis, ms.T_fm, ms.T_mf);
        if (pr.updateDst) {
            float *rd = L + dm.off;
            error_jacobian<false>(rd, pdm, 1.0f, md.depMask[dstFrame], gdm, dm.dims - 6, md.jointType, md.jointAxis, md.T_fm, md.T_mf);
            for (int i = 0; i < dm.dims; ++i) rd[i] = -rd[i];
        }
        const float we = pr.weight * err;
        float fold = we;
        if (pr.updateContact) {
            // the prior's three unknowns lie behind the model blocks; the back substitution (:732) takes the last columns first and
            // these touch no other prior's, so their solution needs this prior's block U = (w g) g^T, b = (w err) g only
            // (:104-108,128-129,151-157), and each solved t_k takes t_k (w J_i) g_k out of the model rows (:173-186)
            const float wg0 = pr.weight * g_sf.x, wg1 = pr.weight * g_sf.y, wg2 = pr.weight * g_sf.z;
            float b0 = we * g_sf.x, b1 = we * g_sf.y, b2 = we * g_sf.z;
            if (b2 != 0.0f) {
                t2 = b2 / (wg2 * g_sf.z);
                const float u02 = wg0 * g_sf.z, u12 = wg1 * g_sf.z;
                if (u02 != 0.0f) b0 -= t2 * u02;
                if (u12 != 0.0f) b1 -= t2 * u12;
            }
            if (b1 != 0.0f) {
                t1 = b1 / (wg1 * g_sf.y);
                const float u01 = wg0 * g_sf.y;
                if (u01 != 0.0f) b0 -= t1 * u01;
            }
            if (b0 != 0.0f) t0 = b0 / (wg0 * g_sf.x);
            fold = ((we - t2 * wg2) - t1 * wg1) - t0 * wg0;
        }
        sc[0] = fold; sc[1] = pr.weight; sc[2] = (pr.updateSrc || pr.updateDst) ? 1.0f : 0.0f;
    } while (0);
    if (pr.updateContact) {     // updatePriorParams with -solution, then back onto the source SDF's surface (:207-221)
        const SdfDev &s = ms.sdfs[pr.srcSdf];
        f3 cp;
        cp.x = pr.contactPoint[0] + (-t0); cp.y = pr.contactPoint[1] + (-t1); cp.z = pr.contactPoint[2] + (-t2);
        const f3 est = project_to_sdf_surface<CNT_PRIOR>(ms.grids, s, grid_coords(s, cp), 1e-9f, 100);
        prp->contactPoint[0] = s.res * est.x + s.ox; prp->contactPoint[1] = s.res * est.y + s.oy; prp->contactPoint[2] = s.res * est.z + s.oz;
    }
}

// iteration: >= 0 the Gauss-Newton iteration; -1 / -2 import / export of a batch's poses; -3 the iteration whose number the
// previous solve left in *a.itCounter (the replayed launches of a captured iteration all carry the same argument)
__global__ void __launch_bounds__(64 * DARTK_MAX_MODELS)
multi_solve_kernel(const MultiSolveArgs *__restrict__ ap, int iteration, int fkFloats) {
    extern __shared__ float lds[];
    const MultiSolveArgs &a = *ap;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;              // wave w assembles and updates model w; wave 0 solves
    const int frame = blockIdx.x;                   // frame of a batch (0 otherwise)
    if (iteration == -3) iteration = *a.itCounter;
    const size_t so = (size_t)frame * a.fb.sys, co = (size_t)frame * a.fb.counts;
    const int D = a.D;
    const int ldU = (D + 1) | 1;
    float *U = lds;                     // [D][ldU] upper triangle of the global system
    float *bv = U + D * ldU;            // [D]
    float *J = bv + D;                  // a prior's Jacobian row over the global unknowns [D] (wave 0)
    float *eJ = J + D * ldU;            // its scalars
    float *ra = eJ + D + (size_t)wave * (64 + fkFloats);   // per wave: reduced articulation of its model [<= 64]
    float *sT = ra + 64;                // per wave: forward-kinematics scratch [fkFloats]
    const bool mine = wave < a.nModels;
    const MultiSolveModel &m = a.m[mine ? wave : 0];
    ModelView v = m.v;
    select_frame(v, frame);
    if (iteration < 0) {        // batch only: import (-1) / export (-2) the caller's poses
        if (!mine) return;
        float *T_io = a.T_mcs + ((size_t)frame * a.nModels + wave) * 12;
        if (iteration == -2) {
            if (lane < 12) T_io[lane] = v.T_mc[lane];
            return;
        }
        const FkTables tab = fk_prefetch(v);
        const RedTables rtab = red_prefetch(m, v);
        if (lane == 0) {
            float T[12], Tinv[12];
            for (int i = 0; i < 12; ++i) T[i] = T_io[i];
            se3_invert(T, Tinv);
            for (int i = 0; i < 12; ++i) { v.T_mc[i] = T[i]; v.T_cm[i] = Tinv[i]; }
        }
        if (lane < m.redArt) ra[lane] = m.redArtPtr[(size_t)frame * a.redStride + lane];
        wave_sync_lds();
        project_and_fk(m, v, ra, sT, tab, rtab);
        return;
    }
#ifdef DARTK_STAMPS
    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    // what the update at the end needs and the solution does not change: in flight from here on
    const FkTables tab = fk_prefetch(v);
    const RedTables rtab = red_prefetch(m, v);
    float T_old[12];
    if (mine && lane == 0)
        for (int i = 0; i < 12; ++i) T_old[i] = v.T_mc[i];
    float redOld = 0.0f, redLo = 0.0f, redHi = 0.0f;
    if (mine && lane < m.redArt) { redOld = m.redArtPtr[(size_t)frame * a.redStride + lane]; redLo = m.redMin[lane]; redHi = m.redMax[lane]; }
    for (int i = threadIdx.x; i < D * ldU; i += blockDim.x) U[i] = 0.0f;
    if (threadIdx.x < D) bv[threadIdx.x] = 0.0f;
    __syncthreads();
    // Assembly of the models' diagonal blocks (optimizer.cpp:640-713). The reference unpacks term after term into the model's JTJ /
    // eJ (Optimizer::unpack, :50-70), adds the regularisation and copies the upper triangle into the global system. Here a lane owns
    // an entry and adds the terms' values in that order in a register -- the same sums -- so that every load of every term is in
    // flight at once instead of one dependent round trip per term, and the models' blocks are assembled side by side, a wave each
    // (the kernel's duration is its chain of latencies).
    if (mine) {
        const int dims = m.dims, off = m.off, dC = dims - 6;
        constexpr int kTerms = 3 + DARTK_MAX_MODELS;
        const float *sys[kTerms];
        float lam[kTerms];
        sys[0] = m.sysA ? m.sysA + so : nullptr; lam[0] = a.lambdaA;
        sys[1] = m.sysB ? m.sysB + so : nullptr; lam[1] = a.lambdaB;
        sys[2] = m.sysC ? m.sysC + so : nullptr; lam[2] = m.lamSelf;          // articulation-only system: the bottom-right block
#pragma unroll
        for (int d = 0; d < DARTK_MAX_MODELS; ++d) {
            const bool on = d < a.nModels && m.sysI[d];
            sys[3 + d] = on ? m.sysI[d] + so : nullptr; lam[3 + d] = m.lamI[d];
        }
        const int S = sys_size(dims);
        if (a.iterInfo && lane == 0) {      // e is assigned per term, the last one stays (SURVEY A-8)
            float eA = 0.0f, eB = 0.0f, eI = 0.0f;
            if (sys[0]) eA = lam[0] * sys[0][S - 1];
            if (sys[1]) eB = lam[1] * sys[1][S - 1];
            if (sys[2]) eI = lam[2] * sys[2][sys_size(dC) - 1];
#pragma unroll
            for (int d = 0; d < DARTK_MAX_MODELS; ++d) if (sys[3 + d]) eI = lam[3 + d] * sys[3 + d][S - 1];
            float *r = a.iterInfo + (size_t)frame * a.infoStride + ((size_t)iteration * a.nModels + wave) * 8;
            r[0] = eA; r[1] = eB; r[2] = eI; r[3] = m.countA ? (float)m.countA[co] : 0.0f; r[4] = m.countB ? (float)m.countB[co] : 0.0f;
            r[5] = r[6] = r[7] = 0.0f;
        }
        // entries [0, dims*dims): the block's (i, j), upper triangle only; [dims*dims, dims*dims + dims): J^T e
        for (int idx = lane; idx < dims * dims + dims; idx += 64) {
            const bool rhs = idx >= dims * dims;
            const int i = rhs ? idx - dims * dims : idx / dims, j = rhs ? 0 : idx - i * dims;
            if (!rhs && j < i) continue;
            float val[kTerms];
#pragma unroll
            for (int t = 0; t < kTerms; ++t) {
                val[t] = 0.0f;
                if (!sys[t]) continue;
                if (t == 2) {
                    if (i < 6 || (!rhs && j < 6)) continue;
                    val[t] = rhs ? sys[t][i - 6] : sys[t][dC + (((j - 6) * (j - 5)) >> 1) + (i - 6)];
                } else
                    val[t] = rhs ? sys[t][i] : sys[t][dims + ((j * (j + 1)) >> 1) + i];
            }
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < kTerms; ++t) {
                if (!sys[t] || (t == 2 && (i < 6 || (!rhs && j < 6)))) continue;
                float tt = 0.0f;
                tt += lam[t] * val[t];
                acc += tt;
            }
            if (rhs) { bv[off + i] = acc; continue; }
            if (i == j) {       // regularisation, LM scaling (optimizer.cpp:695-700)
                acc += m.reg * 1.0f;
                acc += m.regScaled * acc;
            }
            if (m.damping) acc += m.damping[idx];                           // JTJ += *dampingMatrices[m]  (:702-703)
            U[(off + i) * ldU + off + j] = acc;
        }
    }
    __syncthreads();
#ifdef DARTK_STAMPS
    const unsigned long long stamp1 = __builtin_amdgcn_s_memrealtime();
#endif
    if (wave == 0) {
        // priors[p]->computeContribution into the assembled system, in list order (:727-730): J^T e += (w err) J, upper J^T J +=
        // (w J_r) J_c over the source block, the destination block and their cross block (contact_prior.cpp:128-157,187-204)
        for (int p = 0; p < a.nPriors; ++p) {
            float *L = J, *sc = eJ;
            if (lane == 0) multi_eval_contact_prior(a, frame, a.priors + p, L, sc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (sc[2] != 0.0f) {
                const float fold = sc[0], w = sc[1];
                for (int idx = lane; idx < D * D; idx += 64) {
                    const int r = idx / D, c = idx - r * D;
                    if (c < r) continue;
                    const float val = (w * L[r]) * L[c];
                    if (val != 0.0f) U[r * ldU + c] += val;
                }
                for (int i = lane; i < D; i += 64) bv[i] += fold * L[i];
            }
            wave_sync_lds();
        }
        if (a.firstSystem && iteration == 0 && frame == 0) {
            for (int idx = lane; idx < D * D; idx += 64) a.firstSystem[idx] = U[(idx / D) * ldU + (idx % D)];
            for (int i = lane; i < D; i += 64) a.firstSystem[D * D + i] = bv[i];
        }
        // paramUpdate = -U^-1 JTe by back substitution on the upper triangle (Eigen triangularView<Upper>().solve, :732:
        // columns last to first, a column is skipped when its rhs entry is exactly 0)
        float x = lane < D ? bv[lane] : 0.0f;       // D <= 64: lane r holds x[r]
        for (int i = D - 1; i >= 0; --i) {
            float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
            if (xi != 0.0f) {
                xi = xi / U[i * ldU + i];
                if (lane == i) x = xi;
                if (lane < i) {
                    const float u = U[lane * ldU + i];
                    if (u != 0.0f) x -= xi * u;
                }
            }
        }
        x = -x;
        if (lane < D) bv[lane] = x;
    }
    __syncthreads();
#ifdef DARTK_STAMPS
    const unsigned long long stamp2 = __builtin_amdgcn_s_memrealtime();
#endif
    // pose updates (:734-755), projection to the full joint vector, forward kinematics (Tracker::updatePose): a wave per model
    if (mine) {
        const int off = m.off;
        if (lane == 0) {
            float t6[6], dT[12], Tn[12], Tinv[12];
            for (int i = 0; i < 6; ++i) t6[i] = bv[off + i];
            se3_exp(t6, dT);
            se3_mul(dT, T_old, Tn);
            se3_invert(Tn, Tinv);
            for (int i = 0; i < 12; ++i) { v.T_mc[i] = Tn[i]; v.T_cm[i] = Tinv[i]; }
        }
        if (lane < m.redArt) {
            const float val = redOld + bv[off + 6 + lane];
            const float c = fminf(fmaxf(redLo, val), redHi);
            m.redArtPtr[(size_t)frame * a.redStride + lane] = c;
            ra[lane] = c;
        }
        wave_sync_lds();
        project_and_fk(m, v, ra, sT, tab, rtab);
    }
    // this iteration's counts have been read (summaries above, the gated reductions before this launch): ready for the next
    if (threadIdx.x < DARTK_MAX_MODELS) {       // (the arrays hold DARTK_MAX_MODELS counters per frame in either form of the loop)
        if (a.zeroA) a.zeroA[co + threadIdx.x] = 0;
        if (a.zeroB) a.zeroB[co + threadIdx.x] = 0;
    }
    if (threadIdx.x == 0 && frame == 0 && a.itCounter) *a.itCounter = iteration + 1;
#ifdef DARTK_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long stamp3 = __builtin_amdgcn_s_memrealtime();
    if (a.iterInfo && threadIdx.x == 0) {      // 10 ns ticks: assembly | priors + back substitution | pose updates + forward kinematics
        float *r = a.iterInfo + (size_t)frame * a.infoStride + ((size_t)iteration * a.nModels) * 8;
        r[5] = (float)(stamp1 - stamp0); r[6] = (float)(stamp2 - stamp1); r[7] = (float)(stamp3 - stamp2);
    }
#endif
}

// per-device buffers of the loop (argument block, per-term systems, small tables)
struct MultiScratch { void *p; size_t bytes; };
static MultiScratch g_multi[16];
static int multi_scratch(size_t bytes, void **out) {
    int dev = 0;
    DARTK_HIP(hipGetDevice(&dev));
    DARTK_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    if (g_multi[dev].bytes < bytes) {
        if (g_multi[dev].p) { DARTK_HIP(hipDeviceSynchronize()); DARTK_HIP(hipFree(g_multi[dev].p)); }
        g_multi[dev].p = nullptr; g_multi[dev].bytes = 0;
        DARTK_HIP(hipMalloc(&g_multi[dev].p, bytes));
        g_multi[dev].bytes = bytes;
    }
    *out = g_multi[dev].p;
    return DARTK_OK;
}

// One call of Optimizer::optimizePoses for nFrames independent frames side by side (nFrames == 1 and d_T_mcs == nullptr: the
// classic single-frame call on the model handles' own poses). Every launch covers all frames (grid.y = frame; the solve: one
// wave per frame); the frames share the models' SDFs and tables and differ in observation maps, poses and observation SDFs.
static int optimize_poses_impl(const dartk_multi_options *o, const dartk_multi_model *mm, int nFrames, const float *d_vert,
                               const float *d_norm, int width, int height, float *d_prediction, float *d_T_mcs, float *d_redArtBatch,
                               float *d_iterInfo, float *d_firstSystem, hipStream_t st, dartk_contact_prior *d_priors = nullptr,
                               int nPriors = 0) {
    const int n = o->nModels;
    const bool batch = d_T_mcs != nullptr;
    const bool useA = o->lambdaObsToMod > 0, useB = o->lambdaModToObs > 0;
    Workspace *ws;
    int rc = get_workspace(&ws);
    if (rc) return rc;

    static thread_local MultiSolveArgs h;
    memset(&h, 0, sizeof(h));
    Model *mods[DARTK_MAX_MODELS];
    static thread_local Model views[DARTK_MAX_MODELS];       // batch: the handles' static tables with per-frame pose / obs-SDF arrays
    int D = 0, maxFrames = 1, redTotal = 0, redOff[DARTK_MAX_MODELS];
    size_t tableFloats = 0, sysFloats = 0, poseFloats = 0, obsFloats = 0;
    size_t poseOff[DARTK_MAX_MODELS], obsOff[DARTK_MAX_MODELS];
    const int nPix = width * height, nPred = o->predWidth * o->predHeight;
    for (int m = 0; m < n; ++m) {
        mods[m] = reinterpret_cast<Model *>(mm[m].model);
        DARTK_REQUIRE(mods[m], "dartk_optimize_poses: null model %d", m);
        const ModelView &v = mods[m]->v;
        const int redArt = mm[m].redType == 0 ? v.nJoints : mm[m].redArtDims;
        DARTK_REQUIRE(mm[m].redType >= 0 && mm[m].redType <= 2, "model %d: reduction type %d", m, mm[m].redType);
        DARTK_REQUIRE(redArt >= 0 && redArt <= v.nJoints, "model %d: %d reduced dimensions for %d joints", m, redArt, v.nJoints);
        DARTK_REQUIRE(mm[m].redType != 1 || mm[m].d_A, "model %d: linear reduction without its matrix", m);
        DARTK_REQUIRE(mm[m].redType != 2 || mm[m].d_map, "model %d: parameter map without its mapping", m);
        DARTK_REQUIRE(redArt == 0 || ((batch || mm[m].d_redArt) && mm[m].h_redMin && mm[m].h_redMax), "model %d: reduced articulation / limits missing", m);
        DARTK_REQUIRE(!useA || batch || mm[m].d_pts, "model %d: association buffer missing", m);
        DARTK_REQUIRE(!useB || mods[m]->obsVoxels > 0, "model %d has no observation SDF", m);
        h.m[m].redType = mm[m].redType; h.m[m].redArt = redArt; h.m[m].dims = 6 + redArt; h.m[m].off = D;
        D += 6 + redArt;
        redOff[m] = redTotal; redTotal += redArt;
        if (v.nFrames > maxFrames) maxFrames = v.nFrames;
        tableFloats += (size_t)v.nJoints + 2 * (size_t)redArt;              // b | min | max (shared by the frames)
        sysFloats += (size_t)(2 + n) * sys_size(6 + redArt) + sys_size(redArt > 0 ? redArt : 1);
        poseOff[m] = poseFloats; poseFloats += 24 + 24 * (size_t)v.nFrames + (size_t)(v.nJoints > 0 ? v.nJoints : 1);
        obsOff[m] = obsFloats; obsFloats += mods[m]->obsVoxels;
    }
    DARTK_REQUIRE(D <= 64, "global system dimension %d exceeds 64 (one lane per unknown in the solve)", D);
    // per-frame tables: observation-SDF poses [n][12] and pair transforms [n][24]; counters [countA | countB]
    const int tabStride = 36 * n;
    FrameBatch fb = {nFrames, DARTK_MAX_MODELS, (int)sysFloats, tabStride, (size_t)nPix, (size_t)nPred, 0, (size_t)nPix};
    size_t traceOff[DARTK_MAX_MODELS], traceFloats = 0;         // every model's slice of a frame's ray-cast trace buffer
    for (int m = 0; m < n; ++m) { traceOff[m] = traceFloats; traceFloats += ray_trace_floats(o->predWidth, o->predHeight, mods[m]->v.nSdfs); }
    fb.trace = traceFloats;
    if (!batch) fb = FrameBatch{1, 0, 0, 0, 0, 0, 0, 0};

    // one allocation: [args | countA | countB | shared tables | per-frame tables | per-frame systems | the two launch tables |
    // the models' ray-cast tables] and, for a batch, [poses | observation SDFs | distance-transform scratch | prediction maps |
    // association lists]
    const size_t argBytes = (sizeof(MultiSolveArgs) + 255) / 256 * 256;
    const size_t cntInts = 2 * (size_t)DARTK_MAX_MODELS * nFrames + 4;      // + the iteration counter
    size_t floatsSmall = tableFloats + (size_t)nFrames * (tabStride + sysFloats);
    size_t total = argBytes + sizeof(int) * cntInts + sizeof(float) * floatsSmall + 1024;
    const size_t o_union = (total + 255) / 256 * 256;          // two UnionTable blocks (the single-frame loop's shared launches)
    total = o_union + 2 * ((sizeof(UnionTable) + 255) / 256 * 256);
    // the models' ray-cast tables of the current pose ([transforms | tile rectangles], 28 nSdfs + 1 words each), written by the
    // first shared launch of an iteration for the replay that follows it
    const size_t o_rtab = total;
    size_t rtabFloats = 0, rtabOff[DARTK_MAX_MODELS];
    for (int m = 0; m < n; ++m) { rtabOff[m] = rtabFloats; rtabFloats += 28 * (size_t)mods[m]->v.nSdfs + 1; }
    total = o_rtab + (sizeof(float) * rtabFloats + 255) / 256 * 256;
    size_t o_pose = 0, o_obs = 0, o_tmp = 0, o_pred = 0, o_pts = 0;
    if (batch) {
        auto take = [&](size_t bytes) { size_t at = (total + 255) / 256 * 256; total = at + bytes; return at; };
        o_pose = take(sizeof(float) * poseFloats * nFrames);
        o_obs = take(sizeof(float) * obsFloats * nFrames);
        o_tmp = take(sizeof(float) * obsFloats * nFrames);
        o_pred = take(sizeof(float) * 4 * (size_t)nPred * nFrames);
        o_pts = useA ? take(sizeof(dartk_da_point) * (size_t)nPix * nFrames * n) : 0;
    }
    void *base = nullptr;
    rc = multi_scratch(total, &base);
    if (rc) return rc;
    MultiSolveArgs *d_args = reinterpret_cast<MultiSolveArgs *>(base);
    int *d_countA = reinterpret_cast<int *>((char *)base + argBytes);
    int *d_countB = d_countA + (size_t)DARTK_MAX_MODELS * nFrames;
    int *d_itCounter = d_countB + (size_t)DARTK_MAX_MODELS * nFrames;
    float *fl = reinterpret_cast<float *>(d_itCounter + 4);
    if (batch) {
        for (int m = 0; m < n; ++m) {       // model m's frame records lie back to back: [frame][T_mc|T_cm|T_mf|T_fm|joints]
            views[m] = *mods[m];
            ModelView &v = views[m].v;
            const int nF = v.nFrames;
            const size_t rec = 24 + 24 * (size_t)nF + (size_t)(v.nJoints > 0 ? v.nJoints : 1);
            float *p = reinterpret_cast<float *>((char *)base + o_pose) + poseOff[m] * nFrames;
            v.T_mc = p; v.T_cm = p + 12; v.T_mf = p + 24; v.T_fm = p + 24 + 12 * nF; v.joints = p + 24 + 24 * nF;
            v.poseStride = (int)rec;
            v.obsData = reinterpret_cast<float *>((char *)base + o_obs) + obsOff[m] * nFrames;
            v.obsStride = (int)mods[m]->obsVoxels;
            views[m].dtTmp = reinterpret_cast<float *>((char *)base + o_tmp) + obsOff[m] * nFrames;
            mods[m] = &views[m];
        }
        d_prediction = reinterpret_cast<float *>((char *)base + o_pred);
    }
    std::vector<float> tables;
    float *T_obs[DARTK_MAX_MODELS], *T_pair[DARTK_MAX_MODELS];
    float *sysA[DARTK_MAX_MODELS], *sysB[DARTK_MAX_MODELS], *sysC[DARTK_MAX_MODELS], *sysI[DARTK_MAX_MODELS][DARTK_MAX_MODELS];
    dartk_da_point *ptsList[DARTK_MAX_MODELS];
    for (int m = 0; m < n; ++m) {
        const ModelView &v = mods[m]->v;
        MultiSolveModel &s = h.m[m];
        const int redArt = s.redArt;
        s.v = v;
        s.A = mm[m].d_A; s.map = mm[m].d_map;
        s.redArtPtr = batch ? d_redArtBatch + redOff[m] : mm[m].d_redArt;
        s.b = fl + tables.size();
        for (int f = 0; f < v.nJoints; ++f) tables.push_back(mm[m].h_b ? mm[m].h_b[f] : 0.0f);
        s.redMin = fl + tables.size();
        for (int r = 0; r < redArt; ++r) tables.push_back(mm[m].h_redMin[r]);
        s.redMax = fl + tables.size();
        for (int r = 0; r < redArt; ++r) tables.push_back(mm[m].h_redMax[r]);
        s.reg = mm[m].regularization; s.regScaled = mm[m].regularizationScaled;
        s.damping = mm[m].d_damping;
        s.countA = useA ? d_countA + m : nullptr;
        s.countB = useB ? d_countB + m : nullptr;
        ptsList[m] = batch ? reinterpret_cast<dartk_da_point *>((char *)base + o_pts) + (size_t)m * nPix * nFrames : mm[m].d_pts;
    }
    float *tabBase = fl + tableFloats;                       // frame 0's [T_obs | T_pair] table
    for (int m = 0; m < n; ++m) { T_obs[m] = tabBase + 12 * m; T_pair[m] = tabBase + 12 * n + 24 * m; }
    float *sysBase = tabBase + (size_t)nFrames * tabStride;  // frame 0's systems
    size_t so = 0;
    for (int m = 0; m < n; ++m) {
        MultiSolveModel &s = h.m[m];
        const int S = sys_size(s.dims);
        sysA[m] = sysBase + so; so += S;
        sysB[m] = sysBase + so; so += S;
        sysC[m] = sysBase + so; so += sys_size(s.redArt > 0 ? s.redArt : 1);
        for (int d = 0; d < n; ++d) { sysI[m][d] = sysBase + so; so += S; }
        s.sysA = useA ? sysA[m] : nullptr;
        s.sysB = useB ? sysB[m] : nullptr;
        const float lamSelf = o->lambdaIntersection[m + m * n];
        const bool useC = lamSelf > 0 && s.redArt > 0;
        DARTK_REQUIRE(!useC || mm[m].nCloud == 0 || (mm[m].d_cloud && mm[m].d_potential), "model %d: self-intersection needs a cloud and a potential matrix", m);
        s.sysC = useC ? sysC[m] : nullptr;
        s.lamSelf = lamSelf;
        for (int d = 0; d < n; ++d) {
            const float lam = o->lambdaIntersection[m + d * n];
            const bool on = d != m && lam > 0;
            DARTK_REQUIRE(!on || mm[m].nCloud == 0 || mm[m].d_cloud, "model %d: intersection needs a collision cloud", m);
            s.sysI[d] = on ? sysI[m][d] : nullptr;
            s.lamI[d] = lam;
        }
    }
    h.nModels = n; h.D = D; h.numIterations = o->numIterations;
    h.lambdaA = o->lambdaObsToMod; h.lambdaB = o->lambdaModToObs;
    h.iterInfo = d_iterInfo; h.firstSystem = d_firstSystem;
    h.priors = d_priors; h.nPriors = nPriors;
    h.zeroA = useA ? d_countA : nullptr; h.zeroB = useB ? d_countB : nullptr;
    h.itCounter = d_itCounter;
    h.fb = fb; h.redStride = batch ? redTotal : 0; h.infoStride = batch ? (size_t)o->numIterations * n * 8 : 0; h.T_mcs = d_T_mcs;
    // generateObsSdf per model with its start pose (optimizer.cpp:574-580); the full-pose mod->obs wrapper works in that frame.
    // A single frame: the builds of the first two models are the longest part of the call's set-up and depend on nothing else in
    // it (each model has its own volumes), so they go FIRST, on the two side streams, and the caller's stream meanwhile takes the
    // argument blocks, the tables and the valid-pixel lists; the three are joined before the loop.
    const bool sideBuilds = useB && !batch && n > 1 && !prof_enabled() && getenv("DARTK_MULTI_ONE_STREAM") == nullptr;
    auto build_obs_sdf = [&](int m, hipStream_t sb) -> int {
        ProfScope ps(DARTK_PROF_OBS_SDF, sb);
        int rc = gn_obs_sdf_build(d_vert, width, height, mods[m], o->focalLength, sb);
        if (rc == DARTK_OK && hipMemcpy2DAsync(T_obs[m], sizeof(float) * (batch ? tabStride : 12), mods[m]->v.T_mc,
                                               sizeof(float) * (batch ? mods[m]->v.poseStride : 12), 12 * sizeof(float), nFrames,
                                               hipMemcpyDeviceToDevice, sb) != hipSuccess) {
            set_error("copy of the observation-SDF pose failed");
            rc = DARTK_ERR_HIP;
        }
        return rc;
    };
    struct BuildJoin {      // (also on an error return: the caller's stream stays ordered after the side streams)
        Workspace *w; hipStream_t st; bool on;
        void join() {
            if (!on) return;
            on = false;
            (void)hipEventRecord(w->evObs, w->auxStream); (void)hipEventRecord(w->evJoin2, w->auxStream2);
            (void)hipStreamWaitEvent(st, w->evObs, 0); (void)hipStreamWaitEvent(st, w->evJoin2, 0);
        }
        ~BuildJoin() { join(); }
    } buildJoin{ws, st, false};
    if (sideBuilds) {
        DARTK_HIP(hipEventRecord(ws->evFork, st));
        DARTK_HIP(hipStreamWaitEvent(ws->auxStream, ws->evFork, 0));
        DARTK_HIP(hipStreamWaitEvent(ws->auxStream2, ws->evFork, 0));
        buildJoin.on = true;
        for (int m = 0; m < 2; ++m) {
            rc = build_obs_sdf(m, m == 0 ? ws->auxStream : ws->auxStream2);
            if (rc) return rc;
        }
    }
    bool oneCopy = false;
    {   // argument block and tables leave through a pinned slot: the call does not wait for the stream
        // (a single frame: the block, the zeroed counters behind it and the tables behind those lie back to back on the device and
        // leave as ONE copy; a batch's counters are per frame and are cleared by a memset of their own)
        ArgSlot slot;
        const size_t cntBytes = sizeof(int) * cntInts, tabBytes = sizeof(float) * tables.size();
        oneCopy = !batch && argBytes + cntBytes + tabBytes <= arg_slot_bytes();
        rc = arg_ring_acquire(oneCopy ? argBytes + cntBytes + tabBytes : sizeof(h) + tabBytes, &slot);
        if (rc) return rc;
        memcpy(slot.h, &h, sizeof(h));
        if (oneCopy) {
            memset((char *)slot.h + sizeof(h), 0, argBytes - sizeof(h) + cntBytes);
            if (tabBytes) memcpy((char *)slot.h + argBytes + cntBytes, tables.data(), tabBytes);
            rc = arg_ring_upload(d_args, slot, argBytes + cntBytes + tabBytes, st);
            if (rc) return rc;
        } else {
            float *hTab = reinterpret_cast<float *>((char *)slot.h + sizeof(h));
            if (tabBytes) memcpy(hTab, tables.data(), tabBytes);
            DARTK_HIP(hipMemcpyAsync(d_args, slot.h, sizeof(h), hipMemcpyHostToDevice, st));
            if (tabBytes) DARTK_HIP(hipMemcpyAsync(fl, hTab, tabBytes, hipMemcpyHostToDevice, st));
        }
        rc = arg_ring_release(slot, st);
        if (rc) return rc;
    }
    const int ldU = (D + 1) | 1;
    const int fkFloats = 12 * maxFrames + 8;          // forward-kinematics scratch of one wave (= model)
    const size_t ldsSolve = sizeof(float) * ((size_t)2 * D * ldU + 2 * D + (size_t)n * (64 + fkFloats));
    DARTK_REQUIRE(ldsSolve <= 160 * 1024, "LDS requirement exceeds 160 KiB");
    if (ldsSolve > 64 * 1024)
        DARTK_HIP(hipFuncSetAttribute((const void *)multi_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsSolve));
    BatchScope scope(fb);       // every launcher below covers all frames
    if (batch)                  // import the caller's poses: T_mc, projection of the reduced articulation, forward kinematics
        hipLaunchKernelGGL(multi_solve_kernel, dim3(nFrames), dim3(64 * n), ldsSolve, st, d_args, -1, fkFloats);

    // association argument block (pose arrays are updated in place, so once per call)
    void *d_assocArgs = nullptr;
    size_t assocStageBytes = 0;
    if (useA) {
        float thr[DARTK_MAX_MODELS], po[DARTK_MAX_MODELS], pn[3 * DARTK_MAX_MODELS];
        for (int m = 0; m < n; ++m) {
            thr[m] = mm[m].distThreshold; po[m] = mm[m].planeOffset;
            for (int k = 0; k < 3; ++k) pn[3 * m + k] = mm[m].planeNormal[k];
        }
        rc = gn_assoc_multi_prepare(n, mods, thr, po, pn, ptsList, &d_assocArgs, st, &assocStageBytes);
        if (rc) return rc;
    }
    // The observation is fixed for the call: its valid pixels (a few per cent of a 1280x720 map around a hand) are listed once
    // per frame, and every iteration's association walks the lists instead of reading every vertex of every frame again.
    int *d_validIdx = nullptr, *d_nValid = nullptr;
    if (useA && o->numIterations > 0) {
        const int nb = (nPix + 255) / 256;
        rc = ensure_valid_idx(ws, (size_t)nFrames * ((size_t)nPix + 1) + valid_pixel_work_ints(nPix, nFrames));
        if (rc) return rc;
        d_validIdx = ws->validIdx;
        int *d_work = d_validIdx + (size_t)nFrames * nPix;
        d_nValid = d_work + valid_pixel_work_ints(nPix, nFrames);
        (void)nb;
        ProfScope ps(DARTK_PROF_OTHER, st);
        rc = gn_valid_pixels(d_vert, nPix, nFrames, d_work, d_validIdx, d_nValid, st);
        if (rc) return rc;
    }
    if (useB)
        for (int m = sideBuilds ? 2 : 0; m < n; ++m) {
            rc = build_obs_sdf(m, st);
            if (rc) return rc;
        }
    // Three chains per iteration meet only in the solve: ray-cast -> cull -> mod->obs (the longest, on the caller's stream),
    // association -> obs->mod, and the intersection terms; the two shorter ones run on the side streams, each term wrapper
    // writing its partial systems into its chain's lane of the slab buffer.
    struct LaneGuard {
        Workspace *w;
        ~LaneGuard() {
            w->partialLane = 0; w->deferReduce = false; w->measureLanes = false; w->nReduceJobs = 0;
            for (int l = 0; l < kPartialLanes; ++l) w->deferOffset[l] = 0;
        }
    } laneGuard{ws};
    // From the second iteration on the terms' slab reductions are noted and done by ONE launch after the chains have joined
    // (Workspace::deferReduce): seven launches less per iteration in a loop whose pace is the host's launches. The first iteration
    // runs as before and adds up what each chain's slabs need, so that the buffer is grown before anything has to survive in it.
    const bool deferLater = o->numIterations > 1 && getenv("DARTK_NO_DEFER_REDUCE") == nullptr && 2 * n + n + n * n <= kMaxReduceJobs;
    if (useB) { rc = gn_raycast_reserve(traceFloats * nFrames); if (rc) return rc; }
    const bool par = o->numIterations > 0 && getenv("DARTK_MULTI_ONE_STREAM") == nullptr;
    hipStream_t stA = par ? ws->auxStream : st, stC = par ? ws->auxStream2 : st;
    // on an error after the fork the caller's stream must still order after the side streams (they write the slab, trace and
    // system buffers the next call reuses)
    struct JoinGuard {
        Workspace *w; hipStream_t st, a, c; bool on;
        ~JoinGuard() {
            if (!on) return;
            (void)hipEventRecord(w->evJoin, a); (void)hipEventRecord(w->evJoin2, c);
            (void)hipStreamWaitEvent(st, w->evJoin, 0); (void)hipStreamWaitEvent(st, w->evJoin2, 0);
        }
    } joinGuard{ws, st, stA, stC, false};
    // both counter arrays start at zero; every solve launch leaves them cleared for the next iteration
    if (!oneCopy) DARTK_HIP(hipMemsetAsync(d_countA, 0, 2 * (size_t)DARTK_MAX_MODELS * nFrames * sizeof(int), st));
    // ---- the single-frame loop on ONE stream with shared launches (the C2 recipe): per iteration
    //   [trace of every model | association | self- and pairwise intersection terms]  ->  replay + cull
    //   ->  [mod->obs of every model | obs->mod of every model]  ->  one reduction of all partial systems  ->  solve
    // five launches and no event instead of ~15 launches and 7 event calls on three streams: the loop was paced by the host's
    // enqueueing (tools/time_enqueue_multi.py), not by the device. The terms run the same bodies (dartk_ne_bodies.h) on the same
    // data; only the number of workgroups per term -- the partition of the fixed-order sums -- is this path's own.
    UnionTable *d_u1 = reinterpret_cast<UnionTable *>((char *)base + o_union);
    UnionTable *d_u2 = reinterpret_cast<UnionTable *>((char *)base + o_union + (sizeof(UnionTable) + 255) / 256 * 256);
    int u1Blocks = 0, u2Blocks = 0, unionPpt = 1, nUnionJobs = 0;
    bool useTables = false, fuseReplay = false;
    float *d_rtab = reinterpret_cast<float *>((char *)base + o_rtab);
    size_t u1Lds = 0, u2Lds = 0;
    ReduceJob unionJobs[kMaxReduceJobs];
    bool unionPath = !batch && n <= 3 && o->numIterations > 0 && nPriors >= 0 && getenv("DARTK_MULTI_NO_UNION") == nullptr &&
                     getenv("DARTK_MULTI_ONE_STREAM") == nullptr && !prof_enabled();
    if (unionPath) {
        int maxS = 1;
        for (int m = 0; m < n && unionPath; ++m) {
            const ModelView &v = mods[m]->v;
            const MultiSolveModel &sm = h.m[m];
            if (sys_size(sm.dims) > maxS) maxS = sys_size(sm.dims);
            if (useB && (v.nSdfs > NE_BLOCK || v.nSdfs > kMaxRaySdfs)) unionPath = false;
            if (sm.sysC && mm[m].nCloud <= 0) unionPath = false;
            for (int d = 0; d < n; ++d) if (sm.sysI[d] && mm[m].nCloud <= 0) unionPath = false;
        }
        if (maxS > 3 * NE_BLOCK) unionPath = false;
        if (useA && (assocStageBytes == 0 || assocStageBytes > 64 * 1024)) unionPath = false;
        unionPpt = maxS <= NE_BLOCK ? 1 : 3;
    }
    if (unionPath) {
        static thread_local UnionTable t1, t2;
        memset(&t1, 0, sizeof(t1)); memset(&t2, 0, sizeof(t2));
        size_t slabFloats = 0;
        auto seg = [&](UnionTable &t, int &blocks, int kind, int count) -> UnionSeg & {
            UnionSeg &g = t.seg[t.nSeg++];
            g.first = blocks; g.count = count; g.kind = kind;
            blocks += count;
            return g;
        };
        auto slab = [&](UnionSeg &g, int S, float *result, const int *count, int keep, int minCount) {
            g.slabs = reinterpret_cast<float *>(slabFloats);        // offset for now: the buffer is sized below
            unionJobs[nUnionJobs++] = ReduceJob{g.slabs, result, count, g.count, S, keep, minCount};
            slabFloats += ((size_t)g.count * S + 63) / 64 * 64;
        };
        const int tilesPix = (nPix + NE_BLOCK - 1) / NE_BLOCK;
        if (useB) {
            for (int m = 0; m < n; ++m) {      // the traces: the launch's longest workgroups first
                const ModelView &v = mods[m]->v;
                const long long tasks = (long long)((o->predWidth + 7) / 8) * ((o->predHeight + 7) / 8) * v.nSdfs;
                const long long want = (tasks + NE_BLOCK / 64 - 1) / (NE_BLOCK / 64);
                UnionSeg &g = seg(t1, u1Blocks, UK_TRACE, (int)(want < 2048 ? want : 2048));
                g.m = v; g.flx = o->predFl[0]; g.fly = o->predFl[1]; g.ppx = (float)(o->predWidth / 2); g.ppy = (float)(o->predHeight / 2);
                g.width = o->predWidth; g.height = o->predHeight;
                g.trace = reinterpret_cast<float2 *>(ws->rayBuf + traceOff[m]);
                if (u1Lds < sizeof(int) * kRectInts) u1Lds = sizeof(int) * kRectInts;
            }
        }
        if (useA) {
            UnionSeg &g = seg(t1, u1Blocks, UK_ASSOC, tilesPix < 2048 ? tilesPix : 2048);
            g.vert = (const float4 *)d_vert; g.nPix = nPix; g.assocArgs = (const MultiArgs *)d_assocArgs; g.lastElement = d_countA;
            g.validIdx = d_validIdx; g.nValid = d_nValid;
            if (u1Lds < assocStageBytes) u1Lds = assocStageBytes;
        }
        for (int m = 0; m < n; ++m) {
            const MultiSolveModel &sm = h.m[m];
            const ModelView &v = mods[m]->v;
            const int tiles = (mm[m].nCloud + NE_BLOCK - 1) / NE_BLOCK;
            const int blocks = tiles < kMaxPartialBlocks ? tiles : kMaxPartialBlocks;
            if (sm.sysC) {
                UnionSeg &g = seg(t1, u1Blocks, UK_SELF, blocks);
                g.red = sm.redType; g.m = v; g.sites = (const float4 *)mm[m].d_cloud; g.nSites = mm[m].nCloud; g.pot = mm[m].d_potential;
                g.redDims = sm.dims; g.A = mm[m].d_A; g.map = mm[m].d_map;
                slab(g, sys_size(sm.dims - 6), sysC[m], nullptr, 1, 0);
                const size_t l = ne_self_lds_bytes(v, sm.dims, sm.redType);
                if (u1Lds < l) u1Lds = l;
            }
            for (int d = 0; d < n; ++d) {
                if (!sm.sysI[d]) continue;
                UnionSeg &g = seg(t1, u1Blocks, UK_INTER, blocks);
                g.red = sm.redType; g.m = v; g.m2 = mods[d]->v; g.sites = (const float4 *)mm[m].d_cloud; g.nSites = mm[m].nCloud;
                g.redDims = sm.dims; g.A = mm[m].d_A; g.map = mm[m].d_map;
                slab(g, sys_size(sm.dims), sysI[m][d], nullptr, 1, 0);
                const size_t l = ne_inter_lds_bytes(v, mods[d]->v, sm.dims, sm.redType);
                if (u1Lds < l) u1Lds = l;
            }
        }
        // (last in the grid: short workgroups that are done long before the trace's stragglers)
        useTables = useB && t1.nSeg + n <= kMaxUnionSegs && getenv("DARTK_MULTI_NO_RAY_TABLES") == nullptr;
        // with the tables in memory, and few enough of them to stage beside a term's own data, the mod->obs terms form the
        // prediction themselves (each from every model's traces, the first model's term writing the map): no replay launch
        fuseReplay = useTables && rtabFloats <= 4 * NE_BLOCK && getenv("DARTK_MULTI_REPLAY_LAUNCH") == nullptr;
        for (int m = 0; m < n && useTables; ++m) {
            const ModelView &v = mods[m]->v;
            UnionSeg &g = seg(t1, u1Blocks, UK_TABLES, 1);
            g.m = v; g.flx = o->predFl[0]; g.fly = o->predFl[1]; g.ppx = (float)(o->predWidth / 2); g.ppy = (float)(o->predHeight / 2);
            g.width = o->predWidth; g.height = o->predHeight;
            g.slabs = d_rtab + rtabOff[m];
            if (u1Lds < sizeof(float) * (28 * (size_t)v.nSdfs + 1)) u1Lds = sizeof(float) * (28 * (size_t)v.nSdfs + 1);
        }
        for (int m = 0; m < n; ++m) {
            const MultiSolveModel &sm = h.m[m];
            const ModelView &v = mods[m]->v;
            if (useB) {
                const int tiles = (nPred + NE_BLOCK - 1) / NE_BLOCK;
                UnionSeg &g = seg(t2, u2Blocks, UK_M2O, tiles < kMaxPartialBlocks ? tiles : kMaxPartialBlocks);
                g.red = sm.redType; g.m = v; g.pred = (const float4 *)d_prediction; g.nPred = nPred;
                g.T_gc = sm.redType == 0 ? T_obs[m] : v.T_mc;        // (reduced wrappers use the model's current T_mc, modToObs.cu:851-988)
                g.numPredictions = d_countB + m; g.redDims = sm.dims; g.A = mm[m].d_A; g.map = mm[m].d_map;
                slab(g, sys_size(sm.dims), sysB[m], nullptr, 1, 0);
                size_t l = ne_mod2obs_lds_bytes(v, sm.dims, sm.redType);
                if (fuseReplay) {       // raytracePrediction's replay + cullUnobservable (:610-618) inside the term
                    ReplayArgs &r = g.replay;
                    r.flx = o->predFl[0]; r.fly = o->predFl[1]; r.ppx = (float)(o->predWidth / 2); r.ppy = (float)(o->predHeight / 2);
                    r.width = o->predWidth; r.modelNum = m;
                    r.trace = reinterpret_cast<const float2 *>(ws->rayBuf);
                    r.predOut = (float4 *)d_prediction;
                    r.tables = d_rtab; r.tablesStride = 0; r.slabPerTile = 0;
                    ReplayMulti &mu = g.replayMulti;
                    mu.nModels = n;
                    for (int k = 0; k < n; ++k) { mu.nSdfsOf[k] = mods[k]->v.nSdfs; mu.traceOff[k] = traceOff[k]; }
                    mu.obs = (const float4 *)d_vert; mu.ow = width; mu.oh = height;
                    mu.writesMap = m == 0;
                    r.multi = &d_u2->seg[t2.nSeg - 1].replayMulti;        // (the member's address in the device copy of the table)
                    l += sizeof(float) * rtabFloats;
                }
                if (u2Lds < l) u2Lds = l;
            }
            // (diagnostic override, tools/time_single_loop.py sweeps it: 128 / 256 cost 8-25 % of the loop, 1024 is equal, 2048 -4 %)
            static const int o2mWgs = [] {
                const char *e = getenv("DARTK_UNION_O2M_WGS");
                const int v = e ? atoi(e) : 0;
                return (v >= 1 && v <= kMaxPartialBlocks) ? v : 512;
            }();
            if (useA) {
                UnionSeg &g = seg(t2, u2Blocks, UK_O2M, tilesPix < o2mWgs ? tilesPix : o2mWgs);
                g.red = sm.redType; g.m = v; g.pts = ptsList[m]; g.vert = (const float4 *)d_vert; g.nPix = nPix; g.huberDelta = o->huberDelta;
                g.count_ = d_countA + m; g.redDims = sm.dims; g.A = mm[m].d_A; g.map = mm[m].d_map;
                slab(g, sys_size(sm.dims), sysA[m], d_countA + m, 0, 10);
                const size_t l = ne_obs2mod_lds_bytes(v, sm.dims, sm.redType);
                if (u2Lds < l) u2Lds = l;
            }
        }
        rc = ensure_partials(ws, (slabFloats + kPartialLanes - 1) / kPartialLanes + 64);    // (the three lanes are one buffer here)
        if (rc) return rc;
        for (UnionTable *t : {&t1, &t2})
            for (int i = 0; i < t->nSeg; ++i)
                if (t->seg[i].kind >= UK_SELF && t->seg[i].kind != UK_TABLES) t->seg[i].slabs = ws->partials + reinterpret_cast<size_t>(t->seg[i].slabs);
        for (int i = 0; i < nUnionJobs; ++i) unionJobs[i].slabs = ws->partials + reinterpret_cast<size_t>(unionJobs[i].slabs);
        {   // the two tables lie back to back on the device: one copy
            const size_t stride = (sizeof(UnionTable) + 255) / 256 * 256;
            ArgSlot slot;
            rc = arg_ring_acquire(stride + sizeof(UnionTable), &slot);
            if (rc) return rc;
            memcpy(slot.h, &t1, sizeof(UnionTable));
            memcpy((char *)slot.h + stride, &t2, sizeof(UnionTable));
            rc = arg_ring_upload(d_u1, slot, stride + sizeof(UnionTable), st);
            if (rc) return rc;
            rc = arg_ring_release(slot, st);
            if (rc) return rc;
        }
    }
    // the observation SDFs are first read by the mod->obs terms: the shared-launch form joins the builds behind its first launch
    // (trace, association, intersections), which runs beside them; the three-stream form before its loop
    if (!unionPath) buildJoin.join();
    auto run_union_iteration = [&](int it) -> int {
        int rc = DARTK_OK;
        if (u1Blocks > 0) {
            ProfScope ps(DARTK_PROF_RAYCAST, st);
            rc = gn_launch_union(d_u1, u1Blocks, unionPpt, 0, u1Lds, st);
            if (rc) return rc;
        }
        buildJoin.join();       // (first iteration only)
        if (useB && !fuseReplay) {
            const float ppx = (float)(o->predWidth / 2), ppy = (float)(o->predHeight / 2);
            rc = gn_launch_raycast_replay_cull_multi(o->predFl[0], o->predFl[1], ppx, ppy, o->predWidth, o->predHeight, n, mods, traceOff,
                                                     d_prediction, d_vert, width, height, st, nullptr, useTables ? d_rtab : nullptr);
            if (rc) return rc;
        }
        if (u2Blocks > 0) {
            ProfScope ps(DARTK_PROF_NE_MOD2OBS, st);
            rc = gn_launch_union(d_u2, u2Blocks, unionPpt, 1, u2Lds, st);
            if (rc) return rc;
        }
        for (int i = 0; i < nUnionJobs; ++i) ws->reduceJobs[i] = unionJobs[i];
        ws->nReduceJobs = nUnionJobs;
        rc = flush_reduce_jobs(ws, st);
        if (rc) return rc;
        {
            ProfScope ps(DARTK_PROF_SOLVE, st);
            hipLaunchKernelGGL(multi_solve_kernel, dim3(nFrames), dim3(64 * n), ldsSolve, st, d_args, it, fkFloats);
        }
        DARTK_HIP(hipGetLastError());
        return DARTK_OK;
    };
    auto run_iteration = [&](int it, int solveArg) -> int {
        int rc = DARTK_OK;
        if (unionPath) return run_union_iteration(it);
        if (par) {      // the poses of this iteration are final on `st`
            DARTK_HIP(hipEventRecord(ws->evFork, st));
            DARTK_HIP(hipStreamWaitEvent(stA, ws->evFork, 0));
            DARTK_HIP(hipStreamWaitEvent(stC, ws->evFork, 0));
            joinGuard.on = true;
        }
        ws->partialLane = 0;
        if (deferLater && it == 0) { ws->measureLanes = true; for (int l = 0; l < kPartialLanes; ++l) ws->laneNeed[l] = 0; }
        if (useB) {     // raytracePrediction of every model into one map + cullUnobservable (:610-618)
            // the traces of the models are independent: the first runs here, the others on the intersection chain's stream
            // ahead of its terms; the replays follow in model order on this stream (they share the prediction map)
            const float ppx = (float)(o->predWidth / 2), ppy = (float)(o->predHeight / 2);
            for (int m = 0; m < n; ++m) {
                rc = gn_launch_raycast_trace_at(o->predFl[0], o->predFl[1], ppx, ppy, o->predWidth, o->predHeight, mods[m]->v, traceOff[m],
                                                (par && m > 0) ? stC : st);
                if (rc) return rc;
            }
            if (par && n > 1) {
                DARTK_HIP(hipEventRecord(ws->evTrace, stC));
                DARTK_HIP(hipStreamWaitEvent(st, ws->evTrace, 0));      // after this stream's own trace, which takes as long
            }
            // a batch keeps no prediction maps for the caller: the replay leaves each frame's hits as a list (a few percent of the
            // map) and the mod->obs terms walk the list; the hit counter of a frame sits in the last slot of its prediction counters
            int *d_hits = (batch && n < DARTK_MAX_MODELS && gn_replay_hit_lists(n, mods)) ? d_countB + (DARTK_MAX_MODELS - 1) : nullptr;
            rc = gn_launch_raycast_replay_cull_multi(o->predFl[0], o->predFl[1], ppx, ppy, o->predWidth, o->predHeight, n, mods, traceOff,
                                                     d_prediction, d_vert, width, height, st, d_hits);
            if (rc) return rc;
            for (int m = 0; m < n; ++m) {   // reduced wrappers use the model's current T_mc, the full one T_obsSdf_c (modToObs.cu:851-988 vs :713-780)
                const MultiSolveModel &s = h.m[m];
                ProfScope ps(DARTK_PROF_NE_MOD2OBS, st);
                rc = gn_ne_mod2obs_any(s.redType, 6 + mods[m]->v.nJoints, s.dims, d_prediction, o->predWidth, o->predHeight, mods[m],
                                       s.redType == 0 ? T_obs[m] : mods[m]->v.T_mc, mm[m].d_A, mm[m].d_map, sysB[m], d_countB + m, st, d_hits);
                if (rc) return rc;
            }
        }
        ws->partialLane = 1;
        if (useA) {
            {
                ProfScope ps(DARTK_PROF_ASSOC_MULTI, stA);
                rc = gn_assoc_multi_launch(d_vert, nPix, n, d_assocArgs, d_countA, stA, d_validIdx, d_nValid, assocStageBytes, false);
                if (rc) return rc;
            }
            for (int m = 0; m < n; ++m) {
                const MultiSolveModel &s = h.m[m];
                ProfScope ps(DARTK_PROF_FUSED_OBS2MOD, stA);
                rc = gn_launch_ne_obs2mod_dev(s.redType, 6 + mods[m]->v.nJoints, s.dims, d_vert, mods[m], o->huberDelta, ptsList[m],
                                              d_countA + m, nPix, mm[m].d_A, mm[m].d_map, sysA[m], stA);
                if (rc) return rc;
            }
        }
        ws->partialLane = 2;
        for (int m = 0; m < n; ++m) {
            const MultiSolveModel &s = h.m[m];
            const int fullDims = 6 + mods[m]->v.nJoints;
            if (s.sysC) {
                ProfScope ps(DARTK_PROF_NE_SELF, stC);
                rc = gn_ne_self_any(s.redType, mm[m].d_cloud, mm[m].nCloud, fullDims, s.dims, mods[m], mm[m].d_A, mm[m].d_map,
                                    mm[m].d_potential, sysC[m], stC);
                if (rc) return rc;
            }
            for (int d = 0; d < n; ++d) {
                if (!s.sysI[d]) continue;
                ProfScope ps(DARTK_PROF_NE_SELF, stC);
                rc = gn_ne_inter_any(s.redType, mm[m].d_cloud, mm[m].nCloud, fullDims, s.dims, mods[m], mods[d], mm[m].d_A, mm[m].d_map,
                                     T_pair[m], sysI[m][d], stC);
                if (rc) return rc;
            }
        }
        ws->partialLane = 0;
        if (par) {
            DARTK_HIP(hipEventRecord(ws->evJoin, stA));
            DARTK_HIP(hipEventRecord(ws->evJoin2, stC));
            DARTK_HIP(hipStreamWaitEvent(st, ws->evJoin, 0));
            DARTK_HIP(hipStreamWaitEvent(st, ws->evJoin2, 0));
            joinGuard.on = false;
        }
        if (ws->deferReduce) { rc = flush_reduce_jobs(ws, st); if (rc) return rc; }
        {
            ProfScope ps(DARTK_PROF_SOLVE, st);
            hipLaunchKernelGGL(multi_solve_kernel, dim3(nFrames), dim3(64 * n), ldsSolve, st, d_args, solveArg, fkFloats);
        }
        DARTK_HIP(hipGetLastError());
        if (deferLater && it == 0) {
            ws->measureLanes = false;
            size_t need = 0;
            for (int l = 0; l < kPartialLanes; ++l) need = ws->laneNeed[l] > need ? ws->laneNeed[l] : need;
            rc = ensure_partials(ws, need);         // (waits for the device if it has to grow the buffer: first call only)
            if (rc) return rc;
            ws->deferReduce = true;
        }
        return DARTK_OK;
    };
    // (Measured and left out: capturing iteration 1 as a HIP graph -- three streams, forks and joins included -- and launching the
    // graph for the remaining iterations. On ROCm 7.2 hipGraphLaunch of the 15-node graph costs the host 70 us, 4.7 us per node, no
    // less than the ~22 direct calls it replaces: 4.1-4.4 k GN it/s against 5.5 k. The solve kernel's iteration == -3 form, which reads
    // the iteration number from device memory, remains from that experiment.)
    for (int it = 0; it < o->numIterations; ++it) {
        rc = run_iteration(it, it);
        if (rc) return rc;
    }
    ws->deferReduce = false;
    if (batch) {                // export the poses
        hipLaunchKernelGGL(multi_solve_kernel, dim3(nFrames), dim3(64 * n), ldsSolve, st, d_args, -2, fkFloats);
        DARTK_HIP(hipGetLastError());
    }
    return DARTK_OK;
}

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_optimize_poses(const dartk_multi_options *o, const dartk_multi_model *mm, const float *d_vert, const float *d_norm,
                         int width, int height, float *d_prediction, float *d_iterInfo, float *d_firstSystem, void *stream) {
    DARTK_REQUIRE(o && mm && d_vert && d_norm, "dartk_optimize_poses: null argument");
    const int n = o->nModels;
    DARTK_REQUIRE(n >= 1 && n <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", n, DARTK_MAX_MODELS);
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    if (o->lambdaModToObs > 0) {
        DARTK_REQUIRE(d_prediction && o->predWidth > 0 && o->predHeight > 0, "mod->obs term needs a prediction buffer");
        // cullUnobservable reads the 2x2 observation pixels at (px*ow/pw, py*oh/ph): the prediction must be coarser than the
        // observation map (the same requirement as dartk_cull_unobservable)
        DARTK_REQUIRE((o->predWidth - 1) * width / o->predWidth + 1 < width && (o->predHeight - 1) * height / o->predHeight + 1 < height,
                      "cullUnobservable reads a 2x2 observation neighbourhood: the prediction (%dx%d) must be smaller than the observation map (%dx%d)",
                      o->predWidth, o->predHeight, width, height);
    }
    return optimize_poses_impl(o, mm, 1, d_vert, d_norm, width, height, d_prediction, nullptr, nullptr, d_iterInfo, d_firstSystem, as_stream(stream));
}

int dartk_optimize_poses_priors(const dartk_multi_options *o, const dartk_multi_model *mm, dartk_contact_prior *d_priors,
                                const dartk_contact_prior *h_priors, int nPriors, const float *d_vert, const float *d_norm, int width,
                                int height, float *d_prediction, float *d_iterInfo, float *d_firstSystem, void *stream) {
    DARTK_REQUIRE(o && mm, "dartk_optimize_poses_priors: null argument");
    DARTK_REQUIRE(nPriors >= 0 && (nPriors == 0 || (d_priors && h_priors)), "bad prior list");
    const int n = o->nModels;
    DARTK_REQUIRE(n >= 1 && n <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", n, DARTK_MAX_MODELS);
    for (int p = 0; p < nPriors; ++p) {
        const dartk_contact_prior &q = h_priors[p];
        DARTK_REQUIRE(q.srcModel >= 0 && q.srcModel < n && q.dstModel >= 0 && q.dstModel < n && mm[q.srcModel].model && mm[q.dstModel].model,
                      "contact prior %d: model index out of range", p);
        DARTK_REQUIRE(q.srcSdf >= 0 && q.srcSdf < reinterpret_cast<Model *>(mm[q.srcModel].model)->v.nSdfs, "contact prior %d: SDF index out of range", p);
        DARTK_REQUIRE(q.srcModel != q.dstModel, "contact prior %d: source and destination model must differ", p);
    }
    hipStream_t st = as_stream(stream);
    if (nPriors) DARTK_HIP(hipMemcpyAsync(d_priors, h_priors, sizeof(dartk_contact_prior) * (size_t)nPriors, hipMemcpyHostToDevice, st));
    DARTK_REQUIRE(d_vert && d_norm, "dartk_optimize_poses_priors: null argument");
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    if (o->lambdaModToObs > 0) {
        DARTK_REQUIRE(d_prediction && o->predWidth > 0 && o->predHeight > 0, "mod->obs term needs a prediction buffer");
        DARTK_REQUIRE((o->predWidth - 1) * width / o->predWidth + 1 < width && (o->predHeight - 1) * height / o->predHeight + 1 < height,
                      "cullUnobservable reads a 2x2 observation neighbourhood: the prediction (%dx%d) must be smaller than the observation map (%dx%d)",
                      o->predWidth, o->predHeight, width, height);
    }
    return optimize_poses_impl(o, mm, 1, d_vert, d_norm, width, height, d_prediction, nullptr, nullptr, d_iterInfo, d_firstSystem, st, d_priors,
                               nPriors);
}

int dartk_optimize_poses_batch(const dartk_multi_options *o, const dartk_multi_model *mm, int nFrames, const float *d_vert,
                               const float *d_norm, int width, int height, float *d_T_mcs, float *d_redArt, float *d_iterInfo,
                               void *stream) {
    DARTK_REQUIRE(o && mm && d_vert && d_norm && d_T_mcs, "dartk_optimize_poses_batch: null argument");
    const int n = o->nModels;
    DARTK_REQUIRE(n >= 1 && n <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", n, DARTK_MAX_MODELS);
    DARTK_REQUIRE(nFrames >= 1 && nFrames <= 65535, "nFrames %d out of range [1,65535]", nFrames);
    DARTK_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 31), "bad map size %dx%d", width, height);
    DARTK_REQUIRE(o->numIterations >= 0, "negative iteration count");
    int redTotal = 0;
    for (int m = 0; m < n; ++m) {
        DARTK_REQUIRE(mm[m].model, "dartk_optimize_poses_batch: null model %d", m);
        redTotal += mm[m].redType == 0 ? reinterpret_cast<Model *>(mm[m].model)->v.nJoints : mm[m].redArtDims;
    }
    DARTK_REQUIRE(redTotal == 0 || d_redArt, "dartk_optimize_poses_batch: reduced articulations missing");
    if (o->lambdaModToObs > 0) {
        DARTK_REQUIRE(o->predWidth > 0 && o->predHeight > 0, "mod->obs term needs a prediction size");
        DARTK_REQUIRE((o->predWidth - 1) * width / o->predWidth + 1 < width && (o->predHeight - 1) * height / o->predHeight + 1 < height,
                      "cullUnobservable reads a 2x2 observation neighbourhood: the prediction (%dx%d) must be smaller than the observation map (%dx%d)",
                      o->predWidth, o->predHeight, width, height);
    }
    // frames in chunks that bound the per-call device memory (association lists, traces, prediction maps: ~50 MB per 1280x720 frame)
    // (DARTK_BATCH_CHUNK is read on every call so that a test can move the chunk seam; anything but a positive number = 64)
    const char *env = getenv("DARTK_BATCH_CHUNK");
    int chunk = env ? atoi(env) : 64;
    if (chunk < 1) chunk = 64;
    hipStream_t st = as_stream(stream);
    const size_t mapFloats = 4 * (size_t)width * height;
    for (int f0 = 0; f0 < nFrames; f0 += chunk) {
        const int nf = nFrames - f0 < chunk ? nFrames - f0 : chunk;
        int rc = optimize_poses_impl(o, mm, nf, d_vert + f0 * mapFloats, d_norm + f0 * mapFloats, width, height, nullptr,
                                     d_T_mcs + (size_t)f0 * n * 12, d_redArt ? d_redArt + (size_t)f0 * redTotal : nullptr,
                                     d_iterInfo ? d_iterInfo + (size_t)f0 * o->numIterations * n * 8 : nullptr, nullptr, st);
        if (rc) return rc;
    }
    return DARTK_OK;
}

}  // extern "C"
