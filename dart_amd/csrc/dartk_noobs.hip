// dartk_noobs.hip -- ContactGrasp refinement: OptimizerNoObs::optimizePoses / getError (optimizer.cpp:783-1076),
// ContactPrior::computeContribution (contact_prior.cpp:8-205, CPU-only in the reference) and the batch of
// independent grasp candidates (BASELINE configs 3/4), device-resident.
//
// The reference evaluates ~2500 contact priors one by one on the host (host SDF reads, sparse coeffRef inserts into a
// 7.5k-dimensional system whose prior rows are identically zero, SURVEY A-7) around three blocking kernel round trips
// per iteration. Here one iteration of ALL candidates of a batch is a handful of launches:
//   noobs_terms_kernel  one launch per term, grid (workgroups of the term, candidate): self-intersection, pairwise
//                       intersection, contact priors; each workgroup -> one partial packed system (deterministic slabs)
//   noobs_solve_kernel  grid (candidate), one wave: fixed-order reduction of the candidate's slabs, assembly in the
//                       reference's order, regularisation + LM damping, the reference's upper-triangular back
//                       substitution (SURVEY A-1), pose update with joint clamps, forward kinematics
// Poses of the batch live in one HBM array; SDFs, clouds, priors and kinematic tables are shared by all candidates.
#include "dartk_internal.h"
#include "dartk_reduce.h"
#include "dartk_kernels.h"
#include "dartk_fk.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace dartk {

#ifndef NOOBS_ROWS
#define NOOBS_ROWS 32        // rows of parked active points per workgroup (<= NE_BLOCK)
#endif
// TERM_PRIORS_UPD: the contact-prior term when some prior has update_contact_point (own instantiation, so that the contact
// point solve and the surface projection cost the ContactGrasp configuration, whose contact points are frozen, nothing)
enum { TERM_SELF = 0, TERM_INTER = 1, TERM_PRIORS = 2, TERM_POINTS = 3, TERM_PRIORS_UPD = 4 };
__host__ __device__ constexpr bool term_contact(int type) { return type == TERM_PRIORS || type == TERM_PRIORS_UPD; }
__host__ __device__ constexpr bool term_weighted(int type) { return term_contact(type) || type == TERM_POINTS; }
enum { NOOBS_USE_CONTACTS = 1, NOOBS_MOVE_CONTACTS = 2 };      // flags of a terms launch
static constexpr int NOOBS_MAX_TERMS = DARTK_MAX_MODELS * DARTK_MAX_MODELS + 2;

// Bounding spheres of one wave's 64 consecutive cloud points (intersection terms), built once at dartk_noobs_create: the clouds are
// static in their SDFs' frames and arrive grouped by SDF and Z-ordered, so a 64-point tile is one compact patch of one link (or the
// tail of one link and the head of the next: two segments). A sphere is given in the frame of its segment's SDF.
struct TileMeta {
    float cA[3], rA, cB[3], rB;     // segment A = lanes [0, split), segment B = lanes [split, 64)
    int gridA, gridB, split;
    int ok;                         // 0: more than two SDFs in the tile (or not grouped): no tile-level test
};

struct NoObsTerm {
    int type, src, dst;
    const TileMeta *tiles;       // [ceil(nPoints / 64)] or null
    int nPoints, nBlocks;        // points (cloud sites / priors) and workgroups (= slabs) of this term per candidate
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
    int uniformRes[DARTK_MAX_MODELS];   // 1 when all SDFs of the model share one resolution
    // pose reductions (pose.h; optimizer.cpp:841-846, 933-954): dims[] above are the REDUCED dimensions 6 + redArt[]; a
    // candidate's pose record carries the reduced articulation behind its transforms (rebase / red_art below)
    int redType[DARTK_MAX_MODELS];      // 0 null, 1 linear, 2 parameter map
    int redArt[DARTK_MAX_MODELS];       // reduced articulation dimensions (the joint count for type 0)
    const float *redA[DARTK_MAX_MODELS];        // [nJoints][redArt] first derivatives (types 1 and 2)
    const int *redMap[DARTK_MAX_MODELS];        // [nJoints] (type 2)
    const float *redB[DARTK_MAX_MODELS];        // [nJoints] offsets
    const float *redMin[DARTK_MAX_MODELS], *redMax[DARTK_MAX_MODELS];     // [redArt]
    int anyReduced, maxFullDims;
    ModelView model[DARTK_MAX_MODELS];   // static tables only; pose pointers are rebased per candidate
    NoObsTerm term[NOOBS_MAX_TERMS];
    const dartk_contact_prior *priors;
    // update_contact_point: every candidate's own contact points [candidate][prior][3] (the reference mutates the prior
    // objects, contact_prior.cpp:212-219; candidates of a batch are independent, so each carries its copy)
    float *contactPts;
    const dartk_point_prior *pointPriors;     // Point3D3DPrior / Point2D3DPrior list, three rows each (TERM_POINTS)
    int nPointPriors;
};

#ifdef DARTK_STAMPS
// diagnostic build only: cycles of thread 0 of every workgroup per phase, summed per term type
__device__ unsigned long long g_noobsStamps[5][8];
#define NOOBS_STAMP(ph) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&g_noobsStamps[TYPE][ph], now_ - stampT); stampT = now_; } } while (0)
#else
#define NOOBS_STAMP(ph) do { } while (0)
#endif

// workgroups (= partial systems) of one term of one candidate. Every workgroup pays a fixed prologue (pose tables, spheres,
// accumulator set-up, slab store), so large batches use few workgroups per term and small ones one per tile.
static int noobs_blocks_per_term(int tiles, int maxBatch) {
    if (tiles < 1) return 1;
    int want = tiles;
    if (maxBatch >= 64) {
        static const char *env = getenv("DARTK_NOOBS_BLOCKS");      // diagnostic override
        // measured optimum on MI355X (tools/sweep_noobs_blocks.sh, C4 scene): 8 at 256 candidates, 4 at 1024, 2 at 4096
        want = env ? atoi(env) : (maxBatch < 128 ? 16 : maxBatch < 512 ? 8 : maxBatch < 2048 ? 4 : 2);
    }
    return want < tiles ? (want > 0 ? want : 1) : tiles;
}

// LDS row length of a term: the packed-system row, and at least the parked state of an active point (8 / 16 floats)
__host__ __device__ inline int noobs_row_len(int type, int d) {
    const int ld = tile_row_len(d, term_weighted(type));
    const int need = term_weighted(type) ? 16 : 8;
    return ld > need ? ld : need;
}

// pose block of one model of one candidate: T_mc | T_cm | joints | T_mf | T_fm
__device__ __forceinline__ ModelView rebase(const ModelView &s, float *pose) {
    ModelView v = s;
    v.T_mc = pose; v.T_cm = pose + 12; v.joints = pose + 24;
    const int nj = s.nJoints > 0 ? s.nJoints : 1;
    v.T_mf = pose + 24 + nj; v.T_fm = v.T_mf + 12 * s.nFrames;
    return v;
}
static int pose_block_floats(const ModelView &m) { return 24 + (m.nJoints > 0 ? m.nJoints : 1) + 24 * m.nFrames; }
// a reduced model's record continues with its reduced articulation [redArt]
__host__ __device__ inline int red_art_offset(const ModelView &m) { return 24 + (m.nJoints > 0 ? m.nJoints : 1) + 24 * m.nFrames; }

// full joint vector from the reduced articulation (pose.cpp:66-73 linear, :91-97 parameter map); lane f = joint f
__device__ __forceinline__ float project_joint(const NoObsProblem *P, int mi, const float *red, int f) {
    if (P->redType[mi] == 2) return red[P->redMap[mi][f]];
    const int nr = P->redArt[mi];
    float full = P->redB[mi][f];
    for (int r = 0; r < nr; ++r) full = full + red[r] * P->redA[mi][r + f * nr];
    return full;
}

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
    const float *jin = joints + (size_t)c * P->jointStride + P->jointOff[mi];
    if (P->redType[mi] == 0) {
        if (lane < m.nJoints) m.joints[lane] = jin[lane];
    } else {        // the caller's array holds the reduced articulation: keep it, project it (Pose::projectReducedToFull)
        float *red = pose + red_art_offset(m);
        if (lane < P->redArt[mi]) red[lane] = jin[lane];
        if (lane < m.nJoints) m.joints[lane] = project_joint(P, mi, jin, lane);
    }
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

// Destination SDFs that a 64-point tile segment can possibly penetrate: its bounding sphere (centre c in the frame of source SDF
// `grid`, radius R) against the sphere of every destination SDF's negative-value box (make_spheres slot 0, destination model frame).
// Wave-uniform arithmetic, once per tile instead of once per lane. Conservative: a point p of the segment has |p - C| <= R (the
// stored R carries a margin far above the rounding of the two transform paths), so |C - c_g| > R + r_g puts every point outside
// sphere g, hence outside the box, hence the exact path would have found no negative value there: the decisions are unchanged.
// T_ds: source model -> destination model (pairwise term) or null (self term, where pot = the potential-intersection row applies).
__device__ __forceinline__ unsigned long long tile_candidates(const float *c, float R, int grid, const int *srcSdfFrame, const float *T_mf_src,
                                                              const float *T_ds, const float *sphDst, int nDst, const int *pot, int nSrc) {
    const f3 cm = se3_apply_w(T_mf_src + 12 * srcSdfFrame[grid], c[0], c[1], c[2], 1.0f);
    f3 cd = cm;
    if (T_ds) cd = se3_apply_w(T_ds, cm.x, cm.y, cm.z, 1.0f);
    unsigned long long mask = 0ull;
    for (int g = 0; g < nDst; ++g) {
        if (pot && !pot[g + grid * nSrc]) continue;
        const float *sp = sphDst + kSphereStride * g;
        const float r2 = sp[3];
        if (!(r2 >= 0.0f)) continue;                    // empty box: nothing to penetrate
        const float dx = cd.x - sp[0], dy = cd.y - sp[1], dz = cd.z - sp[2];
        const float reach = R + sqrtf(r2) + 1e-5f;
        if (dx * dx + dy * dy + dz * dz <= reach * reach) mask |= 1ull << g;
    }
    return mask;
}

// -----------------------------------------------------------------------------------------------------
// one term of one iteration, grid (workgroups of the term, candidate); one instantiation per term type so that the
// register budget of the contact-prior path does not set the occupancy of the intersection paths.
// LDS: per model [T_fm | T_mf] (24*nFrames) | per model spheres (20*nSdfs) | rows | wave counters | potential-intersection matrix (self term)
// -----------------------------------------------------------------------------------------------------
#ifndef NOOBS_ACC_LDS
#define NOOBS_ACC_LDS 1
#endif
#ifndef NOOBS_DEFER_GRAD_SELF
#define NOOBS_DEFER_GRAD_SELF 1
#endif
#ifndef NOOBS_DEFER_GRAD_INTER
#define NOOBS_DEFER_GRAD_INTER 1
#endif
#ifndef NOOBS_WAVES_PRIORS
#define NOOBS_WAVES_PRIORS 4
#endif
#ifndef NOOBS_WAVES_INTER
#define NOOBS_WAVES_INTER 5
#endif
// (the body: workgroup bx of term termIdx; the kernels below are its wrappers)
template <int TPT, int TYPE>
__device__ __forceinline__ void noobs_terms_body(const NoObsProblem *__restrict__ P, float *__restrict__ poses, float *__restrict__ slabs,
                                                 int termIdx, int potLdsInts, int flags, int bx) {
    extern __shared__ float lds[];
#ifdef DARTK_STAMPS
    unsigned long long stampT = clock64();
#endif
    __shared__ float sT_ds[24];
    __shared__ double sErr[NE_BLOCK / 64];
    __shared__ int sOff[DARTK_MAX_MODELS + 1];
    __shared__ int sSphOff[DARTK_MAX_MODELS + 1];
    __shared__ int sSdfOff[DARTK_MAX_MODELS + 1];      // per model: SdfDev headers, then the SDF -> frame table
    const int c = blockIdx.y;
    const NoObsTerm &T = P->term[termIdx];
    const int nModels = P->nModels;
    float *poseC = poses + (size_t)c * P->poseStride;
    // stage every model's frame transforms: model mi at lds + sOff[mi] as [T_fm | T_mf]
    if (threadIdx.x == 0) {
        int o = 0;
        for (int mi = 0; mi < nModels; ++mi) { sOff[mi] = o; o += 24 * P->model[mi].nFrames; }
        sOff[nModels] = o;
        for (int mi = 0; mi < nModels; ++mi) { sSphOff[mi] = o; o += kSphereStride * P->model[mi].nSdfs; }
        sSphOff[nModels] = o;
        for (int mi = 0; mi < nModels; ++mi) { sSdfOff[mi] = o; o += (kSdfWords + 1) * P->model[mi].nSdfs + 24 + 4 * P->model[mi].nJoints; }
        sSdfOff[nModels] = o;
    }
    __syncthreads();
    // SDF headers and SDF -> frame tables: read per (point, SDF) pair in the test loops, from LDS instead of through
    // generic pointers into the arena (vector loads of wave-uniform values)
    for (int mi = 0; mi < nModels; ++mi) {
        const ModelView &mv = P->model[mi];
        int *p = reinterpret_cast<int *>(lds + sSdfOff[mi]);
        const int *srcW = reinterpret_cast<const int *>(mv.sdfs);
        const int nw = kSdfWords * mv.nSdfs;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) p[i] = srcW[i];
        for (int i = threadIdx.x; i < mv.nSdfs; i += blockDim.x) p[nw + i] = mv.sdfFrame[i];
        const float *pose = poseC + P->poseOff[mi];         // the candidate's [T_mc | T_cm] of this model
        float *pp = lds + sSdfOff[mi] + (kSdfWords + 1) * mv.nSdfs;
        for (int i = threadIdx.x; i < 24; i += blockDim.x) pp[i] = pose[i];
        // joint types and axes: the Jacobian rows index them per lane (error_jacobian_walk)
        int *jt = reinterpret_cast<int *>(pp + 24);
        float *ja = pp + 24 + mv.nJoints;
        for (int i = threadIdx.x; i < mv.nJoints; i += blockDim.x) jt[i] = mv.jointType[i];
        for (int i = threadIdx.x; i < 3 * mv.nJoints; i += blockDim.x) ja[i] = mv.jointAxis[i];
    }
    for (int mi = 0; mi < nModels; ++mi) {
        float *p = lds + sOff[mi];
        const ModelView mv = rebase(P->model[mi], poseC + P->poseOff[mi]);
        const int n = 12 * mv.nFrames;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { p[i] = mv.T_fm[i]; p[n + i] = mv.T_mf[i]; }
    }
    // The offsets come out of LDS, i.e. in vector registers even where every lane reads the same one: the intersection terms index
    // them with the term's (wave-uniform) models, so they are moved to scalar registers there -- a dozen vector registers that the
    // test loops otherwise spill. The prior terms index them with each lane's own prior's models.
    constexpr bool UNI = (TYPE == TERM_SELF || TYPE == TERM_INTER);
#define uniOff(x) (UNI ? __builtin_amdgcn_readfirstlane(x) : (x))
#define sTf(mi) (lds + uniOff(sOff[mi]))
#define sSph(mi) (lds + uniOff(sSphOff[mi]))
#define sSdf(mi) (reinterpret_cast<const SdfDev *>(lds + uniOff(sSdfOff[mi])))
#define sFrm(mi, nSdfs_) (reinterpret_cast<const int *>(lds + uniOff(sSdfOff[mi])) + kSdfWords * (nSdfs_))
#define sPoseT(mi, nSdfs_) (lds + uniOff(sSdfOff[mi]) + (kSdfWords + 1) * (nSdfs_))      /* T_mc at +0, T_cm at +12 */
#define sJType(mi, nSdfs_) (reinterpret_cast<const int *>(sPoseT(mi, nSdfs_) + 24))
#define sJAxis(mi, nSdfs_, nJoints_) (sPoseT(mi, nSdfs_) + 24 + (nJoints_))
    __syncthreads();
    // bounding spheres of every SDF's keep boxes in its model's frame (see make_spheres)
    for (int mi = 0; mi < nModels; ++mi) {
        const ModelView &mv = P->model[mi];
        for (int g = threadIdx.x; g < mv.nSdfs; g += blockDim.x)
            make_spheres(mv.sdfs[g], sTf(mi) + 12 * mv.nFrames + 12 * mv.sdfFrame[g], sSph(mi) + kSphereStride * g);
    }
    const int d = T.dims;
    const int ld = noobs_row_len(TYPE, d);
    // TPT == 0: the outer product runs on the matrix cores (systems of at most 32 dimensions, MfmaAcc), else on 4x4 VALU tiles; the
    // intersection terms keep the matrix-core block in LDS between flushes (mfma_lds_floats ahead of the row buffer)
    constexpr bool ACC_LDS = NOOBS_ACC_LDS && TPT == 0 && (TYPE == TERM_SELF || TYPE == TERM_INTER);
    float *accLds = lds + __builtin_amdgcn_readfirstlane(sSdfOff[nModels]);
    float *rows = accLds + (ACC_LDS ? mfma_lds_floats(T.dims) : 0);
    int *waveCnt = reinterpret_cast<int *>(rows + NOOBS_ROWS * ld);
    int *sPot = waveCnt + 8;            // the self term's potential-intersection matrix (when it fits)
    // reduced poses: scratch for the full Jacobian of a parked point before its reduction (one row per parked point)
    const int fullLd = P->maxFullDims | 1;
    float *sFullJ = reinterpret_cast<float *>(sPot + (TYPE == TERM_SELF ? potLdsInts : 0));
    ModelView src = rebase(P->model[T.src], poseC + P->poseOff[T.src]);
    ModelView dst = rebase(P->model[T.dst], poseC + P->poseOff[T.dst]);
    src.sdfs = sSdf(T.src); src.sdfFrame = sFrm(T.src, src.nSdfs);
    dst.sdfs = sSdf(T.dst); dst.sdfFrame = sFrm(T.dst, dst.nSdfs);
    const int *potPtr = T.pot;
    if (TYPE == TERM_SELF) {
        const int nn = src.nSdfs * src.nSdfs;
        if (nn <= potLdsInts) {
            for (int i = threadIdx.x; i < nn; i += blockDim.x) sPot[i] = T.pot[i];
            potPtr = sPot;
        }
    }
    if (TYPE == TERM_INTER && threadIdx.x == 0) {      // T_ds = T_mc(dst) * T_cm(src), T_sd = inverse (optimizer.cpp:396-397)
        float A[12], B[12], T_ds[12], T_sd[12];
        for (int i = 0; i < 12; ++i) { A[i] = dst.T_mc[i]; B[i] = src.T_cm[i]; }
        se3_mul(A, B, T_ds);
        se3_invert(T_ds, T_sd);
        for (int i = 0; i < 12; ++i) { sT_ds[i] = T_ds[i]; sT_ds[12 + i] = T_sd[i]; }
    }
    __syncthreads();

    // Intersection terms: which destination SDFs each of this workgroup's 64-point tiles can reach at all (tile_candidates), for
    // ALL its tiles at once, one (tile, segment) per thread -- wave-uniform arithmetic costs a SIMD as much as per-lane arithmetic,
    // so the tile test only pays when 64 tiles share one instruction stream. Most tiles of a hand's cloud are nowhere near the
    // other model (or near a link they may collide with): their waves then skip the point transform and the per-SDF walk.
    [[maybe_unused]] unsigned long long *sCull = nullptr;
    [[maybe_unused]] int *sCullSplit = nullptr;
    [[maybe_unused]] int tileIter = 0;
    if constexpr (TYPE == TERM_SELF || TYPE == TERM_INTER) {
        const ModelView &dm = (TYPE == TERM_SELF) ? src : dst;
        if (T.tiles && dm.nSdfs <= 64) {
            const int nT = (T.nPoints + NE_BLOCK - 1) / NE_BLOCK;
            const int mine = (bx < nT) ? (nT - bx + T.nBlocks - 1) / T.nBlocks : 0;     // tiles of this workgroup
            float *endF = sFullJ + ((P->anyReduced) ? NOOBS_ROWS * fullLd : 0);
            sCull = reinterpret_cast<unsigned long long *>(lds + ((((int)(endF - lds)) + 1) & ~1));
            sCullSplit = reinterpret_cast<int *>(sCull + 4 * mine);
            const int nWaveTiles = (T.nPoints + 63) / 64;
            const float *T_ds = (TYPE == TERM_INTER) ? sT_ds : nullptr;
            const int *pot = (TYPE == TERM_SELF) ? potPtr : nullptr;
            const float *sph = (TYPE == TERM_SELF) ? sSph(T.src) : sSph(T.dst);
            for (int w = threadIdx.x; w < 4 * mine; w += blockDim.x) {
                const int seg = w & 1, k = w >> 1;
                const int wt = (bx + (k >> 1) * T.nBlocks) * (NE_BLOCK / 64) + (k & 1);
                unsigned long long mask = ~0ull;
                int split = -1;
                if (wt < nWaveTiles) {
                    const TileMeta tm = T.tiles[wt];
                    if (tm.ok) {
                        split = tm.split;
                        if (seg == 0) mask = tile_candidates(tm.cA, tm.rA, tm.gridA, src.sdfFrame, sTf(T.src) + 12 * src.nFrames, T_ds, sph, dm.nSdfs, pot, src.nSdfs);
                        else mask = (tm.split < 64) ? tile_candidates(tm.cB, tm.rB, tm.gridB, src.sdfFrame, sTf(T.src) + 12 * src.nFrames, T_ds, sph,
                                                                       dm.nSdfs, pot, src.nSdfs) : 0ull;
                    }
                }
                sCull[w] = mask;
                if (seg == 0) sCullSplit[k] = split;
            }
            __syncthreads();
        }
    }
    NOOBS_STAMP(0);
    std::conditional_t<TPT == 0, MfmaAcc<NE_BLOCK, ACC_LDS>, TileAcc<NE_BLOCK, (TPT > 0 ? TPT : 1)>> acc;
    if constexpr (TPT == 0) acc.cLds = accLds;
    acc.init(d);
    if constexpr (TPT == 0) acc.dims = d;
    NOOBS_STAMP(1);
    constexpr bool WEIGHTED = term_weighted(TYPE);
    constexpr bool DEFER_GRAD = (TYPE == TERM_SELF && NOOBS_DEFER_GRAD_SELF) || (TYPE == TERM_INTER && NOOBS_DEFER_GRAD_INTER);
    constexpr bool UPD = TYPE == TERM_PRIORS_UPD;
    const int lOff = tile_l_off(d, WEIGHTED);
    double errInactive = 0.0;       // contact priors: _error of priors that contribute no row
    // rows of the parked points [0, n): Jacobians by the first n threads, then every thread adds its packed entries
    auto flush = [&](int n) {
        __syncthreads();
        NOOBS_STAMP(3);
        // 4 neighbouring lanes share a row: every entry of the row is written by exactly one of them (error_jacobian_walk4), the
        // parked state is read by all four before any of them writes (same wave: LDS accesses complete in program order)
        static_assert(4 * NOOBS_ROWS <= NE_BLOCK, "4 lanes per pending row");
        const int rowIdx = threadIdx.x >> 2, q = threadIdx.x & 3;
        if (rowIdx < n) {
            float *row = rows + rowIdx * ld;
            f3 pm, gs;
            pm.x = row[0]; pm.y = row[1]; pm.z = row[2]; gs.x = row[3]; gs.y = row[4]; gs.z = row[5];
            const float residual = row[6];
            const int packed = WEIGHTED ? __float_as_int(row[7]) : (__float_as_int(row[7]) & 0xff);
            if constexpr (DEFER_GRAD) {
                // intersections park the hit's grid coordinates (in gs): three of the row's lanes look up one central difference each
                // (sdf_gradient's six lookups, intersection.cu:117-131 / 206-222), the lookups of all pending rows in flight together
                // instead of one divergent gather chain per tile with a hit; then the rotation into the source model's frame
                const int hit = (__float_as_int(row[7]) >> 8) & 0xff;
                const ModelView &hm = (TYPE == TERM_SELF) ? src : dst;
                const SdfDev &sd = hm.sdfs[hit];
                float comp = 0.0f;
                if (q < 3) {
                    const float *data = hm.grids + sd.dataOff;
                    f3 qp = gs, qm = gs;
                    if (q == 0) { qp.x = gs.x + 1.0f; qm.x = gs.x - 1.0f; }
                    else if (q == 1) { qp.y = gs.y + 1.0f; qm.y = gs.y - 1.0f; }
                    else { qp.z = gs.z + 1.0f; qm.z = gs.z - 1.0f; }
                    const float fp = sdf_lookup<CNT_INTER>(data, sd.dimx, sd.dimy, sd.nbx, sd.nby, qp);
                    const float fm = sdf_lookup<CNT_INTER>(data, sd.dimx, sd.dimy, sd.nbx, sd.nby, qm);
                    comp = 0.5f * (fp - fm);
                }
                const int l0 = (int)(threadIdx.x & 63u) & ~3;
                f3 gf; gf.x = __shfl(comp, l0); gf.y = __shfl(comp, l0 + 1); gf.z = __shfl(comp, l0 + 2);
                const f3 gd = se3_rotate(sTf(TYPE == TERM_SELF ? T.src : T.dst) + 12 * hm.nFrames + 12 * hm.sdfFrame[hit], gf);
                gs = (TYPE == TERM_SELF) ? gd : se3_rotate(sT_ds + 12, gd);
            }
            if constexpr (WEIGHTED) {
                // subJ = [ dot(grad_sm, J3D_src) | -dot(grad_dm, J3D_dst) ] placed at the models' global offsets (:110-126)
                f3 pdm, gdm;
                pdm.x = row[8]; pdm.y = row[9]; pdm.z = row[10]; gdm.x = row[11]; gdm.y = row[12]; gdm.z = row[13];
                const float weight = row[14], E = row[15];
                __builtin_amdgcn_wave_barrier();
                const int srcFrame = packed & 0xff, dstFrame = (packed >> 8) & 0xff, pSrc = (packed >> 16) & 0xf, pDst = (packed >> 20) & 0xf;
                const bool upSrc = (packed >> 24) & 1, upDst = (packed >> 25) & 1;
                // the J^T J operand R = [r | E | w*L_0 .. w*L_{d-1} | 0 ..] is written in full below; L starts as zeros (the walks
                // fill the dependent joints only). Lane q owns the entries L[i] with (i & 3) == q, from zeroing to scaling.
                float *L = row + lOff;
                for (int i = 2 + d + q; i < lOff; i += 4) row[i] = 0.0f;
                for (int i = q; i < ld - lOff; i += 4) L[i] = 0.0f;       // (lOff >= 2 + d, tile_l_off)
                const ModelView &ms = P->model[pSrc];
                const ModelView &md = P->model[pDst];
                // The priors index the model's Jacobian by the pose's REDUCED dimensions (contact_prior.cpp:31-32,110-126:
                // subJ(i) = dot(grad, J3D[i]) for i < getReducedDimensions(), J3D being the FULL Jacobian): for a reduced pose
                // that is the first 6 + redArt columns of the full Jacobian, kept as the reference computes it
                if (upSrc)
                    error_jacobian_walk4<false, false>(L + P->off[pSrc], pm, 1.0f, ms.depMask[srcFrame], gs, P->dims[pSrc] - 6,
                                                       sJType(pSrc, ms.nSdfs), sJAxis(pSrc, ms.nSdfs, ms.nJoints), sTf(pSrc),
                                                       sTf(pSrc) + 12 * ms.nFrames, q, P->off[pSrc]);
                int negLo = 0, negHi = 0;       // the destination model's columns enter with the opposite sign (:118-126)
                if (upDst) {
                    float *rd = L + P->off[pDst];
                    error_jacobian_walk4<false, false>(rd, pdm, 1.0f, md.depMask[dstFrame], gdm, P->dims[pDst] - 6,
                                                       sJType(pDst, md.nSdfs), sJAxis(pDst, md.nSdfs, md.nJoints), sTf(pDst),
                                                       sTf(pDst) + 12 * md.nFrames, q, P->off[pDst]);
                    negLo = P->off[pDst]; negHi = negLo + P->dims[pDst];
                }
                if ((d & 3) == q) L[d] = 1.0f;
                // J^T e += (w*err) * J (:128); a prior that also solves for its contact point parks that factor with the
                // contact point's share of the back substitution already taken out (see the evaluation below)
                if (q == 0) {
                    row[0] = (UPD && ((packed >> 26) & 1)) ? residual : weight * residual;
                    row[1] = E;
                }
                for (int i = q; i < d; i += 4) {       // (w*J) J^T   (:129), the sign flip of the destination block in the same pass
                    float v = L[i];
                    if (i >= negLo && i < negHi) { v = -v; L[i] = v; }
                    row[2 + i] = weight * v;
                }
            } else {
                __builtin_amdgcn_wave_barrier();
                for (int i = d + 3 + q; i < ld; i += 4) row[i] = 0.0f;
                const int red = P->redType[T.src];
                if (red == 0) {
                    error_jacobian_walk4<TYPE == TERM_SELF, true>(row + 2, pm, 1.0f, src.depMask[packed], gs, src.nJoints, sJType(T.src, src.nSdfs),
                                                                  sJAxis(T.src, src.nSdfs, src.nJoints), sTf(T.src), sTf(T.src) + 12 * src.nFrames,
                                                                  q, 0);
                } else if (q == 0) {        // normEqns*Reduced / *ParamMap (intersection.cu:147-300, 445-597): full Jacobian, then its reduction
                    float *Jf = sFullJ + rowIdx * fullLd;
                    error_jacobian_walk<TYPE == TERM_SELF>(Jf, pm, 1.0f, src.depMask[packed], gs, src.nJoints, sJType(T.src, src.nSdfs),
                                                           sJAxis(T.src, src.nSdfs, src.nJoints), sTf(T.src), sTf(T.src) + 12 * src.nFrames);
                    const int fullDims = 6 + src.nJoints, redDims = P->dims[T.src];
                    if (red == 1) reduce_jacobian<1>(row + 2, Jf, TYPE == TERM_SELF ? 0 : 6, fullDims, redDims, P->redA[T.src], nullptr);
                    else reduce_jacobian<2>(row + 2, Jf, TYPE == TERM_SELF ? 0 : 6, fullDims, redDims, nullptr, P->redMap[T.src]);
                }
                if (q == 0) {
                    row[0] = residual;
                    row[1] = (float)(0.5 * (double)residual * (double)residual);
                    row[d + 2] = 1.0f;
                }
            }
        }
        __syncthreads();
        NOOBS_STAMP(4);
        if (n) acc.template accumulate<WEIGHTED>(rows, ld, n, lOff);
        __syncthreads();
        NOOBS_STAMP(5);
    };
    int pending = 0;
    const int localBlock = bx;
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
        bool folded = false;                   // the parked residual already is the row's J^T e factor (TERM_PRIORS_UPD)
        // intersection terms: which destination SDFs this wave's 64 points can reach at all (tile_candidates). Most tiles of a
        // hand's cloud are nowhere near the other model (or near a link they may collide with): they cost one scalar-rate sphere
        // test per destination SDF instead of 64 lanes x (transform + per-SDF tests).
        [[maybe_unused]] unsigned long long allowed = ~0ull, tileLoop = ~0ull;
        [[maybe_unused]] bool tileOk = false;
        if constexpr (TYPE == TERM_SELF || TYPE == TERM_INTER) {
            if (sCull) {
                const int k = 2 * tileIter + (int)(threadIdx.x >> 6);      // this wave's tile among the workgroup's
                const int split = sCullSplit[k];
                if (split >= 0) {
                    tileOk = true;
                    const unsigned long long mA = sCull[2 * k], mB = sCull[2 * k + 1];
                    tileLoop = mA | mB;
                    allowed = ((int)(threadIdx.x & 63) < split) ? mA : mB;
                }
            }
            ++tileIter;
        }
        if (index < T.nPoints) {
            if constexpr (TYPE == TERM_SELF) {
                if (allowed != 0ull) {
                    const float4 v = T.cloud[index];
                    const int srcGrid = (int)roundf(v.w);
                    srcFrame = src.sdfFrame[srcGrid];
                    pm = se3_apply_w(sTf(T.src) + 12 * src.nFrames + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
                    const Collision col = tileOk ? first_collision_masked(src, sTf(T.src), pm, sSph(T.src), allowed, tileLoop)
                                                 : first_collision(src, sTf(T.src), pm, potPtr, srcGrid, sSph(T.src));
                    if (col.hit) {
                        if constexpr (DEFER_GRAD) {     // the gradient is looked up when the row is built (flush)
                            gs = col.vg; dstFrame = col.dstGrid;
                        } else {
                            const SdfDev &s = src.sdfs[col.dstGrid];
                            const f3 gf = sdf_gradient<CNT_INTER>(src.grids, s, col.vg);
                            gs = se3_rotate(sTf(T.src) + 12 * src.nFrames + 12 * src.sdfFrame[col.dstGrid], gf);
                        }
                        residual = col.residual;
                        active = true;
                    }
                }
            } else if constexpr (TYPE == TERM_INTER) {
                if (allowed != 0ull) {
                    const float4 v = T.cloud[index];
                    const int srcGrid = (int)roundf(v.w);
                    srcFrame = src.sdfFrame[srcGrid];
                    pm = se3_apply_w(sTf(T.src) + 12 * src.nFrames + 12 * srcFrame, v.x, v.y, v.z, 1.0f);
                    const f3 vdm = se3_apply_w(sT_ds, pm.x, pm.y, pm.z, 1.0f);
                    const Collision col = tileOk ? first_collision_masked(dst, sTf(T.dst), vdm, sSph(T.dst), allowed, tileLoop)
                                                 : first_collision(dst, sTf(T.dst), vdm, nullptr, 0, sSph(T.dst));
                    if (col.hit) {
                        if constexpr (DEFER_GRAD) {
                            gs = col.vg; dstFrame = col.dstGrid;
                        } else {
                            const SdfDev &s = dst.sdfs[col.dstGrid];
                            const f3 gf = sdf_gradient<CNT_INTER>(dst.grids, s, col.vg);
                            const f3 gd = se3_rotate(sTf(T.dst) + 12 * dst.nFrames + 12 * dst.sdfFrame[col.dstGrid], gf);
                            gs = se3_rotate(sT_ds + 12, gd);
                        }
                        residual = col.residual;
                        active = true;
                    }
                }
            } else if constexpr (TYPE == TERM_POINTS) {
                // Point3D3DPrior / Point2D3DPrior::computeContribution (point_3d_3d_prior.cpp:6-69, point_2d_3d_prior.cpp:6-83):
                // row k of a prior's 3 (2) x dims Jacobian is dot(e_k, J3D) resp. dot(projection gradient, J3D); thread = (prior, k)
                const int q = index / 3, comp = index - 3 * q;
                const dartk_point_prior pr = P->pointPriors[q];
                if (pr.weight != 0) {
                    pSrc = pr.srcModel; srcFrame = pr.srcFrame;
                    const ModelView ms = rebase(P->model[pSrc], poseC + P->poseOff[pSrc]);
                    const float *smf = sTf(pSrc) + 12 * ms.nFrames + 12 * srcFrame;
                    float T_cm[12], T_mf[12], T_fc[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) { T_cm[i] = ms.T_cm[i]; T_mf[i] = smf[i]; }
                    se3_mul(T_cm, T_mf, T_fc);                       // MirroredModel::getTransformFrameToCamera
                    const f3 pc = se3_apply_w(T_fc, pr.point_f[0], pr.point_f[1], pr.point_f[2], 1.0f);
                    pm = se3_apply_w(smf, pr.point_f[0], pr.point_f[1], pr.point_f[2], 1.0f);
                    if (pr.kind == 0) {
                        f3 dc; dc.x = pc.x - pr.point_c[0]; dc.y = pc.y - pr.point_c[1]; dc.z = pc.z - pr.point_c[2];
                        const float dist = sqrtf(dot(dc, dc));
                        if (dist != 0) {
                            const f3 dm = se3_rotate(ms.T_mc, dc);
                            gs.x = comp == 0 ? 1.f : 0.f; gs.y = comp == 1 ? 1.f : 0.f; gs.z = comp == 2 ? 1.f : 0.f;
                            residual = comp == 0 ? dm.x : (comp == 1 ? dm.y : dm.z);
                            active = true;
                        }
                    } else if (comp < 2) {
                        const float px = pc.x * pr.focalLength[0] / pc.z + pr.principalPoint[0];
                        const float py = pc.y * pr.focalLength[1] / pc.z + pr.principalPoint[1];
                        const float dx = px - pr.point_c[0], dy = py - pr.point_c[1];
                        if (sqrtf(dx * dx + dy * dy) != 0) {
                            f3 gc;       // d(pixel_k)/d(point_c); dot(g_c, R_cm J3D) == dot(R_mc g_c, J3D)
                            if (comp == 0) { gc.x = pr.focalLength[0] / pc.z; gc.y = 0.f; gc.z = -(pc.x) * pr.focalLength[0] / (pc.z * pc.z); }
                            else { gc.x = 0.f; gc.y = pr.focalLength[1] / pc.z; gc.z = -(pc.y) * pr.focalLength[1] / (pc.z * pc.z); }
                            gs = se3_rotate(ms.T_mc, gc);
                            residual = comp == 0 ? dx : dy;
                            active = true;
                        }
                    }
                    weight = pr.weight; upSrc = true; upDst = false; E = 0.0f;
                }
            } else {
                // ContactPrior::computeContribution, contact_prior.cpp:17-95
                dartk_contact_prior pr = P->priors[index];
                float *cpSlot = nullptr;
                float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;      // UPD: the solution of the prior's own three unknowns
                if constexpr (UPD) {
                    if (pr.updateContact && (flags & NOOBS_USE_CONTACTS)) {
                        cpSlot = P->contactPts + ((size_t)c * P->nPriors + index) * 3;
                        pr.contactPoint[0] = cpSlot[0]; pr.contactPoint[1] = cpSlot[1]; pr.contactPoint[2] = cpSlot[2];
                    }
                }
                const int sign = pr.invert ? -1 : 1;
                E = sign * pr.weight * pr.contactThreshold;
                if (pr.weight != 0) {
                    pSrc = pr.srcModel; pDst = pr.dstModel;
                    const ModelView ms = rebase(P->model[pSrc], poseC + P->poseOff[pSrc]);
                    const ModelView md = rebase(P->model[pDst], poseC + P->poseOff[pDst]);
                    const float *sfm = sTf(pSrc), *smf = sTf(pSrc) + 12 * ms.nFrames;
                    const float *dfm = sTf(pDst), *dmf = sTf(pDst) + 12 * md.nFrames;
                    const SdfDev *mdSdfs = sSdf(pDst);
                    const int *mdFrame = sFrm(pDst, md.nSdfs);
                    srcFrame = sFrm(pSrc, ms.nSdfs)[pr.srcSdf];
                    pm = se3_apply_w(smf + 12 * srcFrame, pr.contactPoint[0], pr.contactPoint[1], pr.contactPoint[2], 1.0f);
                    const f3 cc = se3_apply_w(sPoseT(pSrc, ms.nSdfs) + 12, pm.x, pm.y, pm.z, 1.0f);
                    pdm = se3_apply_w(sPoseT(pDst, md.nSdfs), cc.x, cc.y, cc.z, 1.0f);
                    // nearest destination SDF by raw voxel-unit value among isInBoundsInterp grids (:41-54; the
                    // reference loop bound `<= getNumSdfs()` reads one element past the end -- restated with `<`)
                    // With one resolution for all of the model's SDFs the raw-value arg-min and the metric gate order alike,
                    // so SDFs that cannot come within the contact threshold are skipped (same outcome: either the same
                    // nearest SDF, or every SDF is beyond the threshold and the prior returns sign*w*threshold).
                    const int slot = P->uniformRes[pDst] ? keep_slot(pr.contactThreshold) : -1;
                    int dstSdf = -1;
                    float minV = 3.402823466e+38f;
                    const float *sph = sSph(pDst) + 4 * (slot < 0 ? 4 : slot);
                    for (int ds = 0; ds < md.nSdfs; ++ds) {
                        if (sphere_reject(sph + kSphereStride * ds, pdm)) continue;
                        const SdfDev &s = mdSdfs[ds];
                        const f3 df = se3_apply_w(dfm + 12 * mdFrame[ds], pdm.x, pdm.y, pdm.z, 1.0f);
                        const f3 g = grid_coords(s, df);
                        if (!in_bounds_interp(s, g) || !in_keep_box(s, slot, g)) continue;
                        const float v = sdf_value<CNT_PRIOR>(md.grids, s, g);
                        if (v < minV) { minV = v; dstSdf = ds; }
                    }
                    if (dstSdf >= 0) {
                        dstFrame = mdFrame[dstSdf];
                        const SdfDev &s = mdSdfs[dstSdf];
                        const f3 cdf = se3_apply_w(dfm + 12 * dstFrame, pdm.x, pdm.y, pdm.z, 1.0f);
                        const f3 cg = grid_coords(s, cdf);
                        if (in_bounds_grad_interp(s, cg)) {
                            // (the reference interpolates the SDF again here, contact_prior.cpp:62: same SDF, same coordinates -- the
                            // value the search above found; reusing it drops one of the ~5 trilinear lookups a prior costs)
                            float err = minV * s.res;
                            if (!(fabsf(err) > pr.contactThreshold)) {
                                if ((double)fabsf(err) < 1e-4) E = 0.0f;       // :70-73
                                else {
                                    const f3 g_df = sdf_gradient<CNT_PRIOR>(md.grids, s, cg);
                                    gdm = se3_rotate(dmf + 12 * dstFrame, g_df);
                                    const f3 g_c = se3_rotate(sPoseT(pDst, md.nSdfs) + 12, gdm);
                                    gs = se3_rotate(sPoseT(pSrc, ms.nSdfs), g_c);
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
                                        if constexpr (UPD) {
                                            if (pr.updateContact) {
                                                // The prior's three contact-point unknowns sit behind the model blocks (optimizer.cpp:
                                                // 829-832) and the reference back-substitutes the upper triangle last column first
                                                // (:931): these three columns come before every model column and touch no other
                                                // prior's, so their solution needs only this prior's block U = (w g) g^T, b = (w err) g
                                                // (contact_prior.cpp:104-108,128-129,151-157; the regularisation of :162-170 is applied
                                                // to a copy that is never read again). Each solved unknown t_k then takes
                                                // t_k * U[model row][k] = t_k * (w J_i) g_k out of the model rows' right-hand side
                                                // (:173-186): the factor below. Eigen skips a column whose right-hand side is exactly 0.
                                                const f3 g = se3_rotate(sfm + 12 * srcFrame, gs);
                                                const float we = pr.weight * err;
                                                const float wg0 = pr.weight * g.x, wg1 = pr.weight * g.y, wg2 = pr.weight * g.z;
                                                float b0 = we * g.x, b1 = we * g.y, b2 = we * g.z;
                                                if (b2 != 0.0f) {
                                                    t2 = b2 / (wg2 * g.z);
                                                    const float u02 = wg0 * g.z, u12 = wg1 * g.z;
                                                    if (u02 != 0.0f) b0 -= t2 * u02;
                                                    if (u12 != 0.0f) b1 -= t2 * u12;
                                                }
                                                if (b1 != 0.0f) {
                                                    t1 = b1 / (wg1 * g.y);
                                                    const float u01 = wg0 * g.y;
                                                    if (u01 != 0.0f) b0 -= t1 * u01;
                                                }
                                                if (b0 != 0.0f) t0 = b0 / (wg0 * g.x);
                                                residual = ((we - t2 * wg2) - t1 * wg1) - t0 * wg0;
                                                folded = true;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (!active) errInactive += (double)E;
                if constexpr (UPD) {
                    // ContactPrior::updatePriorParams (contact_prior.cpp:207-221) with this prior's part of paramUpdate = -solution
                    // (optimizer.cpp:931,956-958): every prior that updates its contact point, active or not, steps and is
                    // projected back onto its own SDF's surface
                    if (cpSlot && (flags & NOOBS_MOVE_CONTACTS)) {
                        const ModelView &mo = P->model[pr.srcModel];
                        const SdfDev &s = sSdf(pr.srcModel)[pr.srcSdf];
                        f3 cp;
                        cp.x = pr.contactPoint[0] + (-t0); cp.y = pr.contactPoint[1] + (-t1); cp.z = pr.contactPoint[2] + (-t2);
                        const f3 est = project_to_sdf_surface<CNT_PRIOR>(mo.grids, s, grid_coords(s, cp), 1e-9f, 100);
                        cpSlot[0] = s.res * est.x + s.ox; cpSlot[1] = s.res * est.y + s.oy; cpSlot[2] = s.res * est.z + s.oz;   // getWorldCoords
                    }
                }
            }
        }
        // Active points are rare and scattered over the lanes: park their state (compacted, in evaluation order) in the
        // row buffer and build the Jacobian rows later, densely, one thread per parked point.
        NOOBS_STAMP(2);
        int nActive;
        const int slot = compact_slot<NE_BLOCK>(active, waveCnt, nActive);
        // the row buffer holds NOOBS_ROWS rows (small, so that LDS does not set the occupancy): park what fits, flush, repeat
        for (int base = 0;;) {
            const int room = NOOBS_ROWS - pending;
            const int take = (nActive - base < room) ? nActive - base : room;
            if (active && slot >= base && slot < base + take) {
                float *e = rows + (pending + slot - base) * ld;
                e[0] = pm.x; e[1] = pm.y; e[2] = pm.z; e[3] = gs.x; e[4] = gs.y; e[5] = gs.z; e[6] = residual;
                if constexpr (WEIGHTED) {
                    e[7] = __int_as_float(srcFrame | (dstFrame << 8) | (pSrc << 16) | (pDst << 20) | ((int)upSrc << 24) | ((int)upDst << 25) |
                                          ((int)folded << 26));
                    e[8] = pdm.x; e[9] = pdm.y; e[10] = pdm.z; e[11] = gdm.x; e[12] = gdm.y; e[13] = gdm.z; e[14] = weight; e[15] = E;
                } else {
                    e[7] = __int_as_float(srcFrame | (dstFrame << 8));       // intersections: dstFrame carries the SDF that was hit
                }
            }
            pending += take; base += take;
            if (base >= nActive) break;
            flush(pending); pending = 0;        // full
        }
        __syncthreads();
        NOOBS_STAMP(3);
    }
    flush(pending);
    if constexpr (term_contact(TYPE)) {    // add the _error of the row-less priors to the e slot, fixed order
        double v = errInactive;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sErr[threadIdx.x >> 6] = v;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NE_BLOCK / 64; ++w) tot += sErr[w];
        acc.add_e(d, tot);
    }
    float *slab = slabs + (size_t)c * P->slabStride + T.slabOff + (size_t)localBlock * T.S;
    acc.store(slab, d);
    NOOBS_STAMP(6);
#undef sTf
#undef sSph
#undef uniOff
}

template <int TPT, int TYPE>
__global__ void __launch_bounds__(NE_BLOCK) __attribute__((amdgpu_waves_per_eu(term_weighted(TYPE) ? NOOBS_WAVES_PRIORS : NOOBS_WAVES_INTER)))
noobs_terms_kernel(const NoObsProblem *__restrict__ P, float *__restrict__ poses, float *__restrict__ slabs, int termIdx, int potLdsInts,
                   int flags) {
    noobs_terms_body<TPT, TYPE>(P, poses, slabs, termIdx, potLdsInts, flags, (int)blockIdx.x);
}

// The intersection terms of an iteration (self, hand -> object, object -> hand ...) in ONE launch: workgroup blockIdx.x belongs to
// the term whose range of workgroups holds it. Three launches of 2 x 4096 workgroups each end in a tail where the chip runs half
// empty; one launch of 6 x 4096 has one tail. Terms [firstTerm, firstTerm + nUnion) must be self / pairwise terms of <= 32 dims.
__global__ void __launch_bounds__(NE_BLOCK) __attribute__((amdgpu_waves_per_eu(NOOBS_WAVES_INTER)))
noobs_inter_union_kernel(const NoObsProblem *__restrict__ P, float *__restrict__ poses, float *__restrict__ slabs, int firstTerm, int nUnion,
                         int potLdsInts, int flags) {
    int bx = (int)blockIdx.x, term = firstTerm;
    for (int i = 0; i < nUnion; ++i) {
        const int nb = P->term[firstTerm + i].nBlocks;
        term = firstTerm + i;
        if (bx < nb) break;
        bx -= nb;
    }
    if (P->term[term].type == TERM_SELF) noobs_terms_body<0, TERM_SELF>(P, poses, slabs, term, potLdsInts, flags, bx);
    else noobs_terms_body<0, TERM_INTER>(P, poses, slabs, term, potLdsInts, flags, bx);
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
    const float *damping[DARTK_MAX_MODELS];   // [dims][dims] per model or null (TrackerNoObs::getDampingMatrix)
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
    // JTJ += *dampingMatrices[m] (optimizer.cpp:895-896): onto the model blocks once their intersection terms are in, before
    // the first prior contributes (the term table lists the priors last)
    bool damped = a.energyOnly != 0;
    auto add_damping = [&]() {
        damped = true;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int mi = 0; mi < nModels; ++mi) {
            const float *dm = a.damping[mi];
            if (!dm) continue;
            const int dims = P->dims[mi], off = P->off[mi];
            for (int idx = lane; idx < dims * dims; idx += 64) {
                const int i = idx / dims, j = idx - i * dims;
                if (j >= i) U[(off + i) * ld + off + j] += dm[idx];
            }
        }
    };
    for (int t = 0; t < P->nTerms; ++t) {
        const NoObsTerm &T = P->term[t];
        if (!damped && term_weighted(T.type)) add_damping();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int o = lane; o < T.S; o += 64) sys[o] = reduce_slot(slabC + T.slabOff, T.nBlocks, T.S, o);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int d = T.dims;
        if (term_weighted(T.type)) {
            // priors add into the global upper triangle after the model blocks were scattered (:913-916)
            for (int idx = lane; idx < D * D; idx += 64) {
                const int i = idx / D, j = idx - i * D;
                if (j < i) continue;
                const float v = sys[d + ((j * (j + 1)) >> 1) + i];
                if (v != 0) U[i * ld + j] += v;
            }
            for (int i = lane; i < D; i += 64) b[i] += sys[i];
            if (term_contact(T.type)) ePri = sys[T.S - 1];
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
    if (!damped) add_damping();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nPriorsAll = P->nPriors + P->nPointPriors;     // pe / priors.size(), optimizer.cpp:1068
    const float energy = eInt + (nPriorsAll > 0 ? ePri / (float)nPriorsAll : 0.0f);
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
        if (P->redType[mi] == 0) {
            if (lane < mv.nJoints) {
                const float v = mv.joints[lane] + b[off + 6 + lane];
                mv.joints[lane] = fminf(fmaxf(mv.limits[2 * lane], v), mv.limits[2 * lane + 1]);
            }
        } else {        // reduced articulation with its own limits, then Pose::projectReducedToFull (optimizer.cpp:945-952)
            float *red = poseC + P->poseOff[mi] + red_art_offset(mv);
            if (lane < P->redArt[mi]) {
                const float v = red[lane] + b[off + 6 + lane];
                red[lane] = fminf(fmaxf(P->redMin[mi][lane], v), P->redMax[mi][lane]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < mv.nJoints) mv.joints[lane] = project_joint(P, mi, red, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        fk_wave(mv, 0, sT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// every candidate starts from the priors' own contact points
__global__ void noobs_init_contacts_kernel(const NoObsProblem *__restrict__ P, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * P->nPriors) return;
    const dartk_contact_prior &pr = P->priors[i % P->nPriors];
    float *cp = P->contactPts + (size_t)i * 3;
    cp[0] = pr.contactPoint[0]; cp[1] = pr.contactPoint[1]; cp[2] = pr.contactPoint[2];
}

// export poses of the batch
__global__ void noobs_store_poses_kernel(const NoObsProblem *__restrict__ P, const float *__restrict__ poses, float *__restrict__ T_mcs,
                                         float *__restrict__ joints) {
    const int mi = blockIdx.x, c = blockIdx.y;
    const float *pose = poses + (size_t)c * P->poseStride + P->poseOff[mi];
    const int lane = threadIdx.x;
    if (lane < 12) T_mcs[((size_t)c * P->nModels + mi) * 12 + lane] = pose[lane];
    if (P->redType[mi] == 0) {
        if (lane < P->model[mi].nJoints) joints[(size_t)c * P->jointStride + P->jointOff[mi] + lane] = pose[24 + lane];
    } else if (lane < P->redArt[mi]) {
        joints[(size_t)c * P->jointStride + P->jointOff[mi] + lane] = pose[red_art_offset(P->model[mi]) + lane];
    }
}

struct NoObs {
    NoObsProblem h;            // host copy
    NoObsProblem *d;           // device copy
    dartk_contact_prior *d_priors;
    dartk_point_prior *d_pointPriors;
    float *d_redTables;        // reductions: A | b | min | max per model, then the parameter maps
    TileMeta *d_tileMeta[DARTK_MAX_MODELS];    // per model: bounding spheres of its collision cloud's 64-point tiles (or null)
    int nPriors;
    int maxBatch;
    int ldsModels, slabFloats;     // LDS floats of the staged model tables; slab floats per candidate of the fixed terms
    float *poses, *slabs;
    float *contactPts;         // [maxBatch][nPriors][3] when a prior updates its contact point, else null
    float *d_damping[DARTK_MAX_MODELS];     // damping matrices [dims][dims] (allocated by dartk_noobs_set_damping), null = zero
    bool dampingOn[DARTK_MAX_MODELS];
    bool anyUpdate;
    int termFlags;             // NOOBS_* flags of the next terms launch
    int SMax, maxFrames, potLdsInts;
    size_t ldsTerm[NOOBS_MAX_TERMS], ldsSolve;
    bool lambdaActive[DARTK_MAX_MODELS * DARTK_MAX_MODELS];
};

DARTK_COUNTER_ACCESS(noobs)

// TileMeta of a collision cloud (host copy [n][4], xyz in the owning SDF's frame, w = SDF number): per 64 consecutive points the
// bounding sphere(s) of at most two runs of one SDF each; radii rounded up with a margin (1e-5 relative + 1e-6 m) far above the fp32
// rounding of the device's transforms of the centre and of the points.
static void build_tile_meta(const float *cloud, int n, int nSdfs, std::vector<TileMeta> &out) {
    const int nTiles = (n + 63) / 64;
    out.assign((size_t)nTiles, TileMeta{});
    for (int tIdx = 0; tIdx < nTiles; ++tIdx) {
        TileMeta &tm = out[(size_t)tIdx];
        const int i0 = tIdx * 64, i1 = (i0 + 64 < n) ? i0 + 64 : n;
        tm.ok = 1; tm.split = 64; tm.gridA = tm.gridB = 0;
        int runs = 0, runStart[3] = {0, 0, 0}, runGrid[3] = {0, 0, 0};
        for (int i = i0; i < i1; ++i) {
            const int g = (int)roundf(cloud[4 * i + 3]);
            if (g < 0 || g >= nSdfs) { tm.ok = 0; break; }
            if (runs == 0 || g != runGrid[runs - 1]) {
                if (runs == 2) { tm.ok = 0; break; }
                runStart[runs] = i - i0; runGrid[runs] = g; ++runs;
            }
        }
        if (!tm.ok || runs == 0) { tm.ok = 0; continue; }
        auto sphere = [&](int a, int b, float *c, float &r) {       // points [i0 + a, i0 + b)
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int i = i0 + a; i < i0 + b; ++i)
                for (int k = 0; k < 3; ++k) { const double v = cloud[4 * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
            for (int k = 0; k < 3; ++k) c[k] = (float)(0.5 * (lo[k] + hi[k]));
            double r2 = 0.0;
            for (int i = i0 + a; i < i0 + b; ++i) {
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) { const double dlt = (double)cloud[4 * i + k] - (double)c[k]; d2 += dlt * dlt; }
                if (d2 > r2) r2 = d2;
            }
            r = (float)(sqrt(r2) * (1.0 + 1e-5) + 1e-6);
            if (!(r == r) || !(c[0] == c[0]) || !(c[1] == c[1]) || !(c[2] == c[2])) tm.ok = 0;       // NaN in the cloud: no tile test
        };
        const int end = i1 - i0;
        tm.gridA = runGrid[0];
        if (runs == 1) { sphere(0, end, tm.cA, tm.rA); tm.split = 64; tm.gridB = runGrid[0]; tm.rB = 0.0f; }
        else { tm.split = runStart[1]; tm.gridB = runGrid[1]; sphere(0, tm.split, tm.cA, tm.rA); sphere(tm.split, end, tm.cB, tm.rB); }
    }
}

}  // namespace dartk

using namespace dartk;

template <int TPT, int TYPE>
static int noobs_launch_term(NoObs *t, int term, int batch, hipStream_t st) {
    auto kfn = noobs_terms_kernel<TPT, TYPE>;
    const size_t lds = t->ldsTerm[term];
    if (lds > 64 * 1024) DARTK_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool dbgOcc = getenv("DARTK_DEBUG_OCCUPANCY") != nullptr;      // diagnostic: resident workgroups per CU
    if (dbgOcc) {
        static bool once[5][3];
        if (!once[TYPE][TPT]) {
            once[TYPE][TPT] = true;
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, NE_BLOCK, lds);
            fprintf(stderr, "[dartk] noobs_terms_kernel<%d,%d>: %zu B dynamic LDS, %d workgroups of %d threads per CU\n", TPT, TYPE, lds, nb, NE_BLOCK);
        }
    }
    hipLaunchKernelGGL(kfn, dim3(t->h.term[term].nBlocks, batch), dim3(NE_BLOCK), lds, st, t->d, t->poses, t->slabs, term, t->potLdsInts,
                       t->termFlags);
    return DARTK_OK;
}

template <int TYPE>
static int noobs_launch_term_ppt(NoObs *t, int term, int batch, hipStream_t st) {
    const int tiles = tile_count(t->h.term[term].dims);     // 4x4 tiles of the packed system per workgroup
    const bool noMfma = getenv("DARTK_NOOBS_NO_MFMA") != nullptr;       // diagnostic / test: the VALU tiles for every size
    if (t->h.term[term].dims <= 32 && !noMfma) return noobs_launch_term<0, TYPE>(t, term, batch, st);
    if (tiles <= NE_BLOCK) return noobs_launch_term<1, TYPE>(t, term, batch, st);
    if (tiles <= 2 * NE_BLOCK) return noobs_launch_term<2, TYPE>(t, term, batch, st);
    set_error("term %d: system dimension %d needs more than %d accumulator tiles", term, t->h.term[term].dims, 2 * NE_BLOCK);
    return DARTK_ERR_UNSUPPORTED;
}

// one launch per term (the terms write disjoint slabs; the solve kernel that follows reads them all)
static int noobs_launch_terms(NoObs *t, int batch, hipStream_t st) {
    // the leading run of self / pairwise terms that the matrix-core form covers shares one launch (noobs_inter_union_kernel)
    int nUnion = 0, unionBlocks = 0;
    size_t unionLds = 0;
    if (getenv("DARTK_NOOBS_NO_MFMA") == nullptr && getenv("DARTK_NOOBS_NO_UNION") == nullptr) {
        while (nUnion < t->h.nTerms && (t->h.term[nUnion].type == TERM_SELF || t->h.term[nUnion].type == TERM_INTER) && t->h.term[nUnion].dims <= 32) {
            unionBlocks += t->h.term[nUnion].nBlocks;
            if (t->ldsTerm[nUnion] > unionLds) unionLds = t->ldsTerm[nUnion];
            ++nUnion;
        }
    }
    if (nUnion < 2) nUnion = 0;
    if (nUnion) {
        ProfScope ps(DARTK_PROF_NOOBS_TERMS, st);
        if (unionLds > 64 * 1024)
            DARTK_HIP(hipFuncSetAttribute((const void *)noobs_inter_union_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)unionLds));
        hipLaunchKernelGGL(noobs_inter_union_kernel, dim3(unionBlocks, batch), dim3(NE_BLOCK), unionLds, st, t->d, t->poses, t->slabs, 0, nUnion,
                           t->potLdsInts, t->termFlags);
    }
    for (int i = nUnion; i < t->h.nTerms; ++i) {
        int rc;
        ProfScope ps(term_weighted(t->h.term[i].type) ? DARTK_PROF_NOOBS_PRIORS : DARTK_PROF_NOOBS_TERMS, st);
        switch (t->h.term[i].type) {
        case TERM_SELF: rc = noobs_launch_term_ppt<TERM_SELF>(t, i, batch, st); break;
        case TERM_INTER: rc = noobs_launch_term_ppt<TERM_INTER>(t, i, batch, st); break;
        case TERM_POINTS: rc = noobs_launch_term_ppt<TERM_POINTS>(t, i, batch, st); break;
        case TERM_PRIORS_UPD: rc = noobs_launch_term_ppt<TERM_PRIORS_UPD>(t, i, batch, st); break;
        default: rc = noobs_launch_term_ppt<TERM_PRIORS>(t, i, batch, st); break;
        }
        if (rc) return rc;
    }
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

extern "C" {

#ifdef DARTK_STAMPS
int dartk_debug_noobs_stamps(unsigned long long *out24, int reset) {
    DARTK_HIP(hipDeviceSynchronize());
    DARTK_HIP(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_noobsStamps), sizeof(unsigned long long) * 24));
    if (reset) { unsigned long long z[24] = {0}; DARTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_noobsStamps), z, sizeof(z))); }
    return DARTK_OK;
}
#endif

int dartk_noobs_destroy(dartk_noobs_t h) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    if (!t) return DARTK_OK;
    if (t->d) (void)hipFree(t->d);
    if (t->d_priors) (void)hipFree(t->d_priors);
    if (t->d_pointPriors) (void)hipFree(t->d_pointPriors);
    if (t->d_redTables) (void)hipFree(t->d_redTables);
    for (int m = 0; m < DARTK_MAX_MODELS; ++m)
        if (t->d_tileMeta[m]) (void)hipFree(t->d_tileMeta[m]);
    if (t->contactPts) (void)hipFree(t->contactPts);
    for (int m = 0; m < DARTK_MAX_MODELS; ++m)
        if (t->d_damping[m]) (void)hipFree(t->d_damping[m]);
    if (t->poses) (void)hipFree(t->poses);
    if (t->slabs) (void)hipFree(t->slabs);
    delete t;
    return DARTK_OK;
}

int dartk_noobs_create(int nModels, const dartk_model_t *models, const float *const *d_clouds, const int32_t *nCloudPoints,
                       const int32_t *const *d_potentials, const float *h_lambdaIntersection, const dartk_contact_prior *h_priors,
                       int nPriors, int maxBatch, dartk_noobs_t *out) {
    return dartk_noobs_create_reduced(nModels, models, nullptr, d_clouds, nCloudPoints, d_potentials, h_lambdaIntersection, h_priors, nPriors,
                                      maxBatch, out);
}

int dartk_noobs_create_reduced(int nModels, const dartk_model_t *models, const dartk_reduction *h_red, const float *const *d_clouds,
                               const int32_t *nCloudPoints, const int32_t *const *d_potentials, const float *h_lambdaIntersection,
                               const dartk_contact_prior *h_priors, int nPriors, int maxBatch, dartk_noobs_t *out) {
    DARTK_REQUIRE(models && out && h_lambdaIntersection && nCloudPoints && d_clouds && d_potentials, "dartk_noobs_create: null argument");
    DARTK_REQUIRE(nModels >= 1 && nModels <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", nModels, DARTK_MAX_MODELS);
    DARTK_REQUIRE(nPriors >= 0 && (nPriors == 0 || h_priors), "bad prior list");
    DARTK_REQUIRE(maxBatch >= 1, "maxBatch must be >= 1");
    NoObs *t = new NoObs();
    memset(t, 0, sizeof(NoObs));
    NoObsProblem &P = t->h;
    P.nModels = nModels; P.nPriors = nPriors;
    int D = 0, poseOff = 0, jointOff = 0, maxFrames = 1;
    size_t redFloats = 0, redInts = 0;
    for (int m = 0; m < nModels; ++m) {
        Model *mod = reinterpret_cast<Model *>(models[m]);
        if (!mod) { delete t; set_error("dartk_noobs_create: null model %d", m); return DARTK_ERR_INVALID; }
        P.model[m] = mod->v;
        const int nJ = mod->v.nJoints;
        const int type = h_red ? h_red[m].type : 0;
        const int redArt = type == 0 ? nJ : h_red[m].redArtDims;
        if (type < 0 || type > 2 || redArt < 0 || redArt > nJ || (type != 0 && (!h_red[m].h_A || !h_red[m].h_redMin || !h_red[m].h_redMax)) ||
            (type == 2 && !h_red[m].h_map)) {
            delete t; set_error("dartk_noobs_create: bad pose reduction for model %d", m); return DARTK_ERR_INVALID;
        }
        P.redType[m] = type; P.redArt[m] = redArt;
        if (type != 0) { P.anyReduced = 1; redFloats += (size_t)nJ * redArt + nJ + 2 * (size_t)redArt; redInts += nJ; }
        if (6 + nJ > P.maxFullDims) P.maxFullDims = 6 + nJ;
        P.off[m] = D; P.dims[m] = 6 + redArt; D += P.dims[m];
        P.poseOff[m] = poseOff; poseOff += pose_block_floats(mod->v) + (type != 0 ? redArt : 0);
        P.jointOff[m] = jointOff; jointOff += redArt;
        if (mod->v.nFrames > maxFrames) maxFrames = mod->v.nFrames;
        P.uniformRes[m] = mod->uniformRes ? 1 : 0;
    }
    if (P.anyReduced) {         // one device block: per reduced model A | b | min | max, then the parameter maps
        std::vector<float> tab(redFloats + redInts);
        hipError_t e0 = hipMalloc((void **)&t->d_redTables, sizeof(float) * tab.size());
        if (e0 != hipSuccess) { delete t; set_error("dartk_noobs_create: %s", hipGetErrorString(e0)); return DARTK_ERR_HIP; }
        size_t o = 0;
        for (int m = 0; m < nModels; ++m) {
            if (P.redType[m] == 0) continue;
            const int nJ = P.model[m].nJoints, nr = P.redArt[m];
            P.redA[m] = t->d_redTables + o; memcpy(tab.data() + o, h_red[m].h_A, sizeof(float) * (size_t)nJ * nr); o += (size_t)nJ * nr;
            P.redB[m] = t->d_redTables + o;
            for (int f = 0; f < nJ; ++f) tab[o + f] = h_red[m].h_b ? h_red[m].h_b[f] : 0.0f;
            o += nJ;
            P.redMin[m] = t->d_redTables + o; memcpy(tab.data() + o, h_red[m].h_redMin, sizeof(float) * nr); o += nr;
            P.redMax[m] = t->d_redTables + o; memcpy(tab.data() + o, h_red[m].h_redMax, sizeof(float) * nr); o += nr;
        }
        for (int m = 0; m < nModels; ++m) {
            if (P.redType[m] == 0) continue;
            const int nJ = P.model[m].nJoints;
            P.redMap[m] = reinterpret_cast<const int *>(t->d_redTables + o);
            int *dst = reinterpret_cast<int *>(tab.data() + o);
            for (int f = 0; f < nJ; ++f) {
                const int v = h_red[m].h_map ? h_red[m].h_map[f] : 0;
                if (P.redType[m] == 2 && (v < 0 || v >= P.redArt[m])) { dartk_noobs_destroy(reinterpret_cast<dartk_noobs_t>(t)); set_error("model %d: parameter map entry %d out of range", m, v); return DARTK_ERR_INVALID; }
                dst[f] = v;
            }
            o += nJ;
        }
        e0 = hipMemcpy(t->d_redTables, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice);
        if (e0 != hipSuccess) { dartk_noobs_destroy(reinterpret_cast<dartk_noobs_t>(t)); set_error("dartk_noobs_create: %s", hipGetErrorString(e0)); return DARTK_ERR_HIP; }
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
        if (q.updateContact) t->anyUpdate = true;
    }
    // term table in the reference's evaluation order: per model self-intersection, then the other models; priors last
    int nTerms = 0, slabOff = 0, SMax = 0;
    int ldsModels = 0, potInts = 0;
    for (int m = 0; m < nModels; ++m) {
        ldsModels += 24 * P.model[m].nFrames + kSphereStride * P.model[m].nSdfs + (kSdfWords + 1) * P.model[m].nSdfs + 24 + 4 * P.model[m].nJoints;
        if (P.model[m].nSdfs * P.model[m].nSdfs > potInts) potInts = P.model[m].nSdfs * P.model[m].nSdfs;
    }
    t->potLdsInts = potInts <= 4096 ? potInts : 0;
    auto add = [&](int type, int src, int dst, int nPoints, int dims, const float *cloud, const int *pot) {
        NoObsTerm &T = P.term[nTerms++];
        T.type = type; T.src = src; T.dst = dst; T.nPoints = nPoints; T.dims = dims; T.S = sys_size(dims);
        const int tiles = (nPoints + NE_BLOCK - 1) / NE_BLOCK;
        T.nBlocks = noobs_blocks_per_term(tiles, maxBatch);
        T.slabOff = slabOff; slabOff += T.nBlocks * T.S;
        T.cloud = reinterpret_cast<const float4 *>(cloud); T.pot = pot;
        // tile-level sphere tests pay for the self term, where they also fold the potential-intersection row of the tile's link
        // (measured per launch of 4096 candidates: 416 -> 296 us); the pairwise terms got slower with them (hand -> one-SDF object:
        // the per-lane test is a single sphere already, 433 -> 494 us; object -> hand 293 -> 311 us) and run without
        T.tiles = (type == TERM_SELF) ? t->d_tileMeta[src] : nullptr;
        if (T.S > SMax) SMax = T.S;
        const int ld = noobs_row_len(type, dims);
        size_t cullFloats = (type == TERM_SELF || type == TERM_INTER) ? 2 + 10 * (size_t)((tiles + T.nBlocks - 1) / T.nBlocks) : 0;   // masks + splits
        if ((type == TERM_SELF || type == TERM_INTER) && dims <= 32) cullFloats += mfma_lds_floats(dims);      // the matrix-core block (ACC_LDS)
        t->ldsTerm[nTerms - 1] = sizeof(float) * ((size_t)ldsModels + (size_t)NOOBS_ROWS * ld + 8 + (type == TERM_SELF ? t->potLdsInts : 0) +
                                                  (P.anyReduced && !term_weighted(type) ? (size_t)NOOBS_ROWS * (P.maxFullDims | 1) : 0) + cullFloats);
    };
    for (int m = 0; m < nModels; ++m) {      // tile spheres of every cloud that is the source of a self-intersection term
        const bool used = h_lambdaIntersection[m + m * nModels] > 0 && P.dims[m] > 6;
        if (!used || nCloudPoints[m] <= 0 || !d_clouds[m] || getenv("DARTK_NOOBS_NO_TILE_CULL")) continue;
        std::vector<float> hc((size_t)4 * nCloudPoints[m]);
        std::vector<TileMeta> meta;
        hipError_t ec = hipMemcpy(hc.data(), d_clouds[m], sizeof(float) * hc.size(), hipMemcpyDeviceToHost);
        if (ec == hipSuccess) {
            build_tile_meta(hc.data(), nCloudPoints[m], P.model[m].nSdfs, meta);
            ec = hipMalloc((void **)&t->d_tileMeta[m], sizeof(TileMeta) * meta.size());
            if (ec == hipSuccess) ec = hipMemcpy(t->d_tileMeta[m], meta.data(), sizeof(TileMeta) * meta.size(), hipMemcpyHostToDevice);
        }
        if (ec != hipSuccess) { dartk_noobs_destroy(reinterpret_cast<dartk_noobs_t>(t)); set_error("dartk_noobs_create: %s", hipGetErrorString(ec)); return DARTK_ERR_HIP; }
    }
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
    if (nPriors > 0) add(t->anyUpdate ? TERM_PRIORS_UPD : TERM_PRIORS, 0, 0, nPriors, D, nullptr, nullptr);
    for (int i = 0; i < nModels * nModels; ++i) t->lambdaActive[i] = h_lambdaIntersection[i] > 0;
    P.nTerms = nTerms; P.slabStride = slabOff > 0 ? slabOff : 1;
    t->ldsModels = ldsModels; t->slabFloats = slabOff;
    t->SMax = SMax; t->maxFrames = maxFrames; t->maxBatch = maxBatch; t->nPriors = nPriors;
    size_t ldsTermMax = 0;
    for (int i = 0; i < nTerms; ++i) if (t->ldsTerm[i] > ldsTermMax) ldsTermMax = t->ldsTerm[i];
    const int ldU = (D + 1) | 1;
    const int sysFloats = SMax > 12 * maxFrames ? SMax : 12 * maxFrames;
    t->ldsSolve = sizeof(float) * ((size_t)D * ldU + D + sysFloats + 8);
    if (ldsTermMax > 160 * 1024 || t->ldsSolve > 160 * 1024) { delete t; set_error("LDS requirement exceeds 160 KiB"); return DARTK_ERR_UNSUPPORTED; }
    hipError_t e = hipSuccess;
    if (nPriors > 0) {
        e = hipMalloc((void **)&t->d_priors, sizeof(dartk_contact_prior) * (size_t)nPriors);
        if (e == hipSuccess) e = hipMemcpy(t->d_priors, h_priors, sizeof(dartk_contact_prior) * (size_t)nPriors, hipMemcpyHostToDevice);
    }
    P.priors = t->d_priors;
    if (e == hipSuccess && t->anyUpdate) e = hipMalloc((void **)&t->contactPts, sizeof(float) * 3 * (size_t)nPriors * maxBatch);
    P.contactPts = t->contactPts;
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

int dartk_noobs_set_damping(dartk_noobs_t h, int modelNum, const float *h_matrix, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t, "dartk_noobs_set_damping: null handle");
    DARTK_REQUIRE(modelNum >= 0 && modelNum < t->h.nModels, "model %d out of range [0,%d)", modelNum, t->h.nModels);
    if (!h_matrix) { t->dampingOn[modelNum] = false; return DARTK_OK; }
    const size_t bytes = sizeof(float) * (size_t)t->h.dims[modelNum] * t->h.dims[modelNum];
    if (!t->d_damping[modelNum]) DARTK_HIP(hipMalloc((void **)&t->d_damping[modelNum], bytes));
    // a set-up call: the matrix may still be read by a solve in flight on another stream, and h_matrix is pageable
    DARTK_HIP(hipDeviceSynchronize());
    DARTK_HIP(hipMemcpyAsync(t->d_damping[modelNum], h_matrix, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    DARTK_HIP(hipStreamSynchronize(as_stream(stream)));
    t->dampingOn[modelNum] = true;
    return DARTK_OK;
}

int dartk_noobs_set_priors(dartk_noobs_t h, const dartk_contact_prior *h_priors, int nPriors, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && (h_priors || nPriors == 0), "dartk_noobs_set_priors: null argument");
    DARTK_REQUIRE(nPriors == t->nPriors, "the number of priors (%d) is fixed at creation (%d)", nPriors, t->nPriors);
    for (int p = 0; p < nPriors; ++p)
        DARTK_REQUIRE(!h_priors[p].updateContact || t->anyUpdate,
                      "contact prior %d: update_contact_point switched on after dartk_noobs_create (the term's form is fixed at creation)", p);
    if (nPriors) {
        DARTK_HIP(hipMemcpyAsync(t->d_priors, h_priors, sizeof(dartk_contact_prior) * (size_t)nPriors, hipMemcpyHostToDevice, as_stream(stream)));
        DARTK_HIP(hipStreamSynchronize(as_stream(stream)));
    }
    return DARTK_OK;
}

}  // extern "C" (reopened below)

// the TERM_POINTS term is always the last one: drop it / append it, then re-size what depends on the term table
static int noobs_set_points_term(NoObs *t, int n) {
    NoObsProblem &P = t->h;
    if (P.nTerms > 0 && P.term[P.nTerms - 1].type == TERM_POINTS) --P.nTerms;
    int slabOff = t->slabFloats;
    if (n > 0) {
        NoObsTerm &T = P.term[P.nTerms];
        T.type = TERM_POINTS; T.src = T.dst = 0; T.nPoints = 3 * n; T.dims = P.D; T.S = sys_size(P.D);
        const int tiles = (T.nPoints + NE_BLOCK - 1) / NE_BLOCK;
        T.nBlocks = noobs_blocks_per_term(tiles, t->maxBatch);
        T.slabOff = slabOff; slabOff += T.nBlocks * T.S;
        T.cloud = nullptr; T.pot = nullptr; T.tiles = nullptr;
        t->ldsTerm[P.nTerms] = sizeof(float) * ((size_t)t->ldsModels + (size_t)NOOBS_ROWS * noobs_row_len(TERM_POINTS, P.D) + 8);
        DARTK_REQUIRE(t->ldsTerm[P.nTerms] <= 160 * 1024, "LDS requirement exceeds 160 KiB");
        if (T.S > t->SMax) t->SMax = T.S;
        ++P.nTerms;
    }
    P.nPointPriors = n;
    P.pointPriors = t->d_pointPriors;
    const int newStride = slabOff > 0 ? slabOff : 1;
    if (newStride != P.slabStride) {
        if (t->slabs) DARTK_HIP(hipFree(t->slabs));
        t->slabs = nullptr;
        P.slabStride = newStride;
        DARTK_HIP(hipMalloc((void **)&t->slabs, sizeof(float) * (size_t)P.slabStride * t->maxBatch));
    }
    const int ldU = (P.D + 1) | 1;
    const int sysFloats = t->SMax > 12 * t->maxFrames ? t->SMax : 12 * t->maxFrames;
    t->ldsSolve = sizeof(float) * ((size_t)P.D * ldU + P.D + sysFloats + 8);
    DARTK_REQUIRE(t->ldsSolve <= 160 * 1024, "LDS requirement exceeds 160 KiB");
    DARTK_HIP(hipMemcpy(t->d, &P, sizeof(NoObsProblem), hipMemcpyHostToDevice));
    return DARTK_OK;
}

extern "C" {

int dartk_noobs_set_point_priors(dartk_noobs_t h, const dartk_point_prior *h_priors, int n, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && (h_priors || n == 0) && n >= 0, "dartk_noobs_set_point_priors: bad argument");
    for (int p = 0; p < n; ++p) {
        const dartk_point_prior &q = h_priors[p];
        DARTK_REQUIRE(q.srcModel >= 0 && q.srcModel < t->h.nModels, "point prior %d: model %d out of range", p, q.srcModel);
        DARTK_REQUIRE(q.srcFrame >= 0 && q.srcFrame < t->h.model[q.srcModel].nFrames, "point prior %d: frame %d out of range", p, q.srcFrame);
        DARTK_REQUIRE(q.kind == 0 || q.kind == 1, "point prior %d: kind must be 0 (3D-3D) or 1 (2D-3D)", p);
    }
    DARTK_HIP(hipStreamSynchronize(as_stream(stream)));       // set-up call: the buffers it replaces may still be in use
    if (n != t->h.nPointPriors) {
        DARTK_HIP(hipDeviceSynchronize());
        if (t->d_pointPriors) DARTK_HIP(hipFree(t->d_pointPriors));
        t->d_pointPriors = nullptr;
        if (n > 0) DARTK_HIP(hipMalloc((void **)&t->d_pointPriors, sizeof(dartk_point_prior) * (size_t)n));
        const int rc = noobs_set_points_term(t, n);
        if (rc) return rc;
    }
    if (n > 0) DARTK_HIP(hipMemcpy(t->d_pointPriors, h_priors, sizeof(dartk_point_prior) * (size_t)n, hipMemcpyHostToDevice));
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
    for (int m = 0; m < DARTK_MAX_MODELS; ++m) a.damping[m] = (m < o->nModels && t->dampingOn[m]) ? t->d_damping[m] : nullptr;
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
    if (t->anyUpdate)
        hipLaunchKernelGGL(noobs_init_contacts_kernel, dim3((batch * t->nPriors + 255) / 256), dim3(256), 0, st, t->d, batch);
    NoObsSolveArgs a;
    noobs_fill_args(t, o, a);
    a.energy = d_energy; a.iterInfo = d_iterInfo; a.firstSystem = d_firstSystem;
    for (int it = 0; it < o->numIterations; ++it) {
        t->termFlags = NOOBS_USE_CONTACTS | NOOBS_MOVE_CONTACTS;
        rc = noobs_launch_terms(t, batch, st);
        if (rc) return rc;
        a.iteration = it; a.energyOnly = 0;
        ProfScope ps(DARTK_PROF_NOOBS_SOLVE, st);
        hipLaunchKernelGGL(noobs_solve_kernel, dim3(batch), dim3(64), t->ldsSolve, st, t->d, t->poses, t->slabs, a);
    }
    // getError at the final poses and contact points (optimizer.cpp:980-982)
    t->termFlags = NOOBS_USE_CONTACTS;
    rc = noobs_launch_terms(t, batch, st);
    if (rc) return rc;
    a.energyOnly = 1;
    hipLaunchKernelGGL(noobs_solve_kernel, dim3(batch), dim3(64), t->ldsSolve, st, t->d, t->poses, t->slabs, a);
    hipLaunchKernelGGL(noobs_store_poses_kernel, dim3(t->h.nModels, batch), dim3(64), 0, st, t->d, t->poses, d_T_mcs, d_joints);
    DARTK_HIP(hipGetLastError());
    return DARTK_OK;
}

int dartk_noobs_contact_points(dartk_noobs_t h, int batch, float *d_contactPoints, void *stream) {
    NoObs *t = reinterpret_cast<NoObs *>(h);
    DARTK_REQUIRE(t && d_contactPoints, "dartk_noobs_contact_points: null argument");
    DARTK_REQUIRE(batch >= 1 && batch <= t->maxBatch, "batch %d out of range [1,%d]", batch, t->maxBatch);
    DARTK_REQUIRE(t->anyUpdate, "no contact prior of this tracker updates its contact point");
    DARTK_HIP(hipMemcpyAsync(d_contactPoints, t->contactPts, sizeof(float) * 3 * (size_t)t->nPriors * batch, hipMemcpyDeviceToDevice,
                             as_stream(stream)));
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
    t->termFlags = 0;           // the priors' own contact points
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
