// dartk_kernels.h -- device building blocks shared by the term kernels and by the fused device-resident
// Gauss-Newton loop (association, normal test, LDS staging, pose reductions).
#pragma once

#include "dartk_internal.h"
#include "dartk_reduce.h"

namespace dartk {

static constexpr int NE_BLOCK = 128;   // lanes (= points per tile) of every normal-equation workgroup

// Independent frames processed side by side by one launch (grid.y = frame), dartk_optimize_poses_batch. Strides are what lies
// between two frames in the respective arrays; partial-system slabs need none (frame f's slabs follow frame f-1's:
// (blockIdx.y * gridDim.x + blockIdx.x) * S). The launchers read the calling thread's current batch (one frame, all strides 0,
// unless a batched loop has set it), so every operator-API entry point keeps its single-frame meaning.
struct FrameBatch {
    int n;                  // frames
    int counts;             // ints per frame in the counter arrays (association / prediction counts)
    int sys;                // floats per frame in the table of reduced packed systems
    int tab;                // floats per frame in the table of observation-SDF poses and pair transforms
    size_t vert;            // float4 per frame in the observation vertex / normal maps
    size_t pred;            // float4 per frame in the prediction map
    size_t trace;           // floats per frame in the ray-cast trace buffer
    size_t pts;             // records per frame in every model's association list
};
FrameBatch &cur_batch();    // host: the calling thread's current batch
struct BatchScope {         // sets the current batch for a scope
    FrameBatch saved;
    explicit BatchScope(const FrameBatch &b) : saved(cur_batch()) { cur_batch() = b; }
    ~BatchScope() { cur_batch() = saved; }
};

// ring of (pinned host, device) argument slots, see dartk_model.hip
struct ArgSlot { void *d; void *h; int idx, dev; };
size_t arg_slot_bytes();
int arg_ring_acquire(size_t bytes, ArgSlot *out);
int arg_ring_upload(void *d_dst, const ArgSlot &s, size_t bytes, hipStream_t st);      // slot -> device, by a kernel (never holds the host)
int arg_ring_release(const ArgSlot &s, hipStream_t st);
int ensure_partials(Workspace *w, size_t floats);
void registry_add(Model *m);           // dartk_tables.hip: live model handles by their device arrays
void registry_remove(Model *m);
int noalloc_scratch_resolve();         // dartk_obs2mod.hip: resolves the current device's address of the no-allocation entry point's scratch array (dartk_init)
int get_multi_args_buffer(void **d_ptr, size_t bytes);
int ensure_ray_buf(Workspace *w, size_t floats);
// floats of ray-cast trace records one model can need: {detection t, hit t} per pixel of every 8x8 tile, per SDF
inline size_t ray_trace_floats(int width, int height, int nSdfs) {
    return (size_t)2 * 64 * ((width + 7) / 8) * ((height + 7) / 8) * (size_t)nSdfs;
}
int ensure_valid_idx(Workspace *w, size_t ints);

// kernel-only launchers used by the device-resident Gauss-Newton loops (dartk_gn.hip)
int gn_raycast_reserve(size_t floats);
int gn_launch_raycast_trace_at(float flx, float fly, float ppx, float ppy, int width, int height, const ModelView &m, size_t bufOffset,
                               hipStream_t st);
int gn_launch_raycast_replay_at(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                                float *d_prediction, size_t bufOffset, hipStream_t st);
int gn_launch_raycast_replay_cull_multi(float flx, float fly, float ppx, float ppy, int width, int height, int nModels,
                                        Model *const *mods, const size_t *bufOffsets, float *d_prediction, const float *d_obs, int ow,
                                        int oh, hipStream_t st, int *d_hitCount = nullptr, const float *d_tables = nullptr);
// batches: the replay can leave each frame's hits back to back at the start of its map region (d_hitCount[frame * counts stride]
// of them) instead of the map; gn_ne_mod2obs_any then walks the list (d_nItems = the same counters)
bool gn_replay_hit_lists(int nModels, Model *const *mods);
int gn_launch_raycast(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                      float *d_prediction, hipStream_t st);
int gn_launch_cull(float *d_pred, int pw, int ph, const float *d_obs, int ow, int oh, hipStream_t st);
// per frame the indices of the pixels with a valid vertex (w > 0), in pixel order: the observation is fixed for a whole
// optimisation call, so the association kernels of its iterations walk this list instead of the map
int gn_valid_pixels(const float *d_vert, int nPix, int nFrames, int *d_work, int *d_validIdx, int *d_nValid, hipStream_t st);
// ints of scratch gn_valid_pixels needs at d_work: per-workgroup counts, then one validity bit per pixel (the second pass reads the
// bits the first pass left instead of the vertex map again)
inline size_t valid_pixel_work_ints(int nPix, int nFrames) {
    return (size_t)nFrames * ((nPix + 255) / 256) + 4 + 2 * (size_t)nFrames * ((nPix + 63) / 64);
}
inline unsigned long long *valid_pixel_masks(int *d_work, int nPix, int nFrames) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_work + (size_t)nFrames * ((nPix + 255) / 256));
    return reinterpret_cast<unsigned long long *>((a + 7) & ~(uintptr_t)7);
}
int gn_obs_sdf_build(const float *d_vert, int width, int height, Model *mod, float focalLength, hipStream_t st,
                     float *d_T_gc_out = nullptr /* optional: receives the camera->model transform the splat was made with */);
int gn_launch_raycast_ne_mod2obs(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum, const ModelView &m,
                                 float *d_prediction, const float *d_T_gc, float *slabs, int nBlocks, int *d_num, hipStream_t st,
                                 hipEvent_t beforeNe = nullptr, bool traceDone = false /* the caller has launched the trace on st */,
                                 const float *d_tables = nullptr /* the replay tables of this pose, if an earlier kernel left them */,
                                 float *d_batchTables = nullptr /* a batch: [frames][28 nSdfs + 1] floats this call fills and uses */);
int gn_launch_ne_mod2obs(const float *d_pred, int width, int height, const ModelView &m, const float *d_T_gc,
                         float *slabs, int nBlocks, int *d_num, hipStream_t st);
int gn_launch_ne_self(const float *d_sites, int nSites, const ModelView &m, const int *d_pot, float *slabs, int nBlocks,
                      hipStream_t st);

// whole terms for the device-resident multi-model loop (dartk_multi.hip); red: 0 full pose, 1 linear reduction, 2 parameter map
int gn_launch_ne_obs2mod_dev(int red, int fullDims, int redDims, const float *d_vert, Model *mod, float huberDelta,
                             const dartk_da_point *d_pts, const int *d_count, int maxElements, const float *d_A,
                             const int *d_map, float *d_result, hipStream_t st);
int gn_assoc_multi_entry(const float *d_vert, const float *d_norm, int width, int height, int nModels, const dartk_model_t *models,
                         const float *d_T_mcs, const float *h_distThresholds, const float *h_planeOffsets, const float *h_planeNormals3,
                         dartk_da_point *const *h_pts_table, int32_t *d_last, int32_t *d_dbgDA, float *d_dbgErr, void *stream);
int gn_assoc_multi_prepare(int nModels, Model *const *mods, const float *distThr, const float *planeOffset, const float *pn3,
                           dartk_da_point *const *pts, void **d_args_out, hipStream_t st, size_t *stageBytes = nullptr);
// stageBytes (from the prepare call): LDS for the staged form of the kernel, 0 = read the tables from memory
int gn_assoc_multi_launch(const float *d_vert, int nPix, int nModels, const void *d_args, int *d_last, hipStream_t st,
                          const int *d_validIdx = nullptr, const int *d_nValid = nullptr, size_t stageBytes = 0, bool zeroCounts = true);
int gn_ne_mod2obs_any(int red, int fullDims, int redDims, const float *d_pred, int width, int height, Model *mod,
                      const float *d_T_gc, const float *d_A, const int *d_map, float *d_result, int *d_num, hipStream_t st,
                      const int *d_nItems = nullptr);
int gn_ne_self_any(int red, const float *d_sites, int nSites, int fullDims, int redDims, Model *mod, const float *d_A,
                   const int *d_map, const int *d_pot, float *d_result, hipStream_t st);
int gn_ne_inter_any(int red, const float *d_sites, int nSites, int fullDims, int redDims, Model *src, Model *dst,
                    const float *d_A, const int *d_map, float *d_T_scratch24, float *d_result, hipStream_t st);

// ---- several term kernels in ONE launch (the single-frame multi-model loop, dartk_multi.hip) ---------------------------------
// A union launch is a list of segments; segment i owns the workgroups [first, first + count) of the grid and runs one kernel
// body (dartk_ne_bodies.h, the ray-cast trace) on them with its own arguments, as if it had been launched alone with `count`
// workgroups. The table lives in device memory (written once per call: nothing in it changes between iterations).
struct MultiArgs;
struct ReplayArgs {            // REPLAY variant: the prediction is formed in the kernel from the ray-cast traces
    float flx, fly, ppx, ppy;
    int width, modelNum;       // modelNum: the label's model number (raycast.cu's modelNum argument)
    const float2 *trace;       // [nSdfs][nPix] {detection t, hit t}
    float4 *predOut;
    // optional: this pose's [per-SDF ray transforms (24 nSdfs floats) | tile rectangles (4 nSdfs + 1 ints)] as an earlier kernel of
    // the iteration left them (replay_tables_to_global): one coalesced load instead of recomputing them in every workgroup
    const float *tables;
    int tablesStride;          // floats between two frames' tables (a batch), else 0
    // a batch: a workgroup walks several tiles (the staging of its prologue paid once) yet leaves ONE partial system per TILE, at
    // the tile's index -- the slabs, hence every sum, are those of a launch with one workgroup per tile
    int slabPerTile;
    // REPLAY == 2 (the multi-model single-frame loop): the pixel's prediction is the closest hit over EVERY model of the call, in
    // model order (what raycast_replay_multi_kernel leaves in the map), then cullUnobservable; `tables` holds the models' tables
    // back to back, `trace` is the base of the trace buffer. The per-model part sits in device memory (read through the scalar
    // cache with a uniform index; by value its indexed arrays would put the whole argument block into scratch).
    const struct ReplayMulti *multi;
};
struct ReplayMulti {
    int nModels, nSdfsOf[DARTK_MAX_MODELS];
    unsigned long long traceOff[DARTK_MAX_MODELS];      // each model's slice of the trace buffer (floats)
    const float4 *obs; int ow, oh;      // the observed vertex map of cullUnobservable (modToObs.cu:682-709)
    int writesMap;                      // this term's workgroups leave the map in predOut (one of the models' terms does)
};


enum UnionKind { UK_TRACE = 0, UK_ASSOC = 1, UK_SELF = 2, UK_INTER = 3, UK_M2O = 4, UK_O2M = 5,
                 UK_TABLES = 6 };    // one workgroup: the model's [ray transforms | tile rectangles] of this pose -> slabs (28 nSdfs + 1 words)
struct UnionSeg {
    int first, count, kind, red;        // red: 0 full pose, 1 linear reduction, 2 parameter map (the body's RED)
    ModelView m, m2;                    // the term's model (the source model of an intersection) / the destination model
    float *slabs;                       // the term's partial systems [count][S]
    int redDims; const float *A; const int *map;
    const float4 *sites; int nSites; const int *pot;                                   // intersections
    float flx, fly, ppx, ppy; int width, height; float2 *trace;                       // trace
    const float4 *vert; int nPix; const MultiArgs *assocArgs; int *lastElement; const int *validIdx, *nValid;   // association
    const float4 *pred; int nPred; const float *T_gc; int *numPredictions;            // mod->obs
    const dartk_da_point *pts; float huberDelta; const int *count_;                   // obs->mod (list length on the device)
    ReplayArgs replay;                  // mod->obs with replay.tables set: the term forms the prediction itself (ne_mod2obs_body, REPLAY == 2)
    ReplayMulti replayMulti;            // (replay.multi points at this member of the table in device memory)
};
static constexpr int kMaxUnionSegs = 16;
struct UnionTable { int nSeg, pad; UnionSeg seg[kMaxUnionSegs]; };
// ppt: 1 (every system of the launch has at most NE_BLOCK entries) or 3 (at most 3 NE_BLOCK)
// group: 0 = a table of trace / association / intersection segments, 1 = a table of mod->obs / obs->mod segments
int gn_launch_union(const UnionTable *d_table, int totalBlocks, int ppt, int group, size_t ldsBytes, hipStream_t st);
size_t ne_mod2obs_lds_bytes(const ModelView &m, int d, int red);
size_t ne_obs2mod_lds_bytes(const ModelView &m, int d, int red);
size_t ne_self_lds_bytes(const ModelView &m, int redDims, int red);
size_t ne_inter_lds_bytes(const ModelView &src, const ModelView &dst, int redDims, int red);

// optional per-kernel-class timing with HIP events on the launch stream (dartk_profile_begin / _end)
int prof_push(int cls, hipStream_t st);       // returns an event-pair index or -1 when profiling is off
void prof_pop(int idx, hipStream_t st);
bool prof_enabled();
struct ProfScope {
    hipStream_t st; int idx;
    ProfScope(int cls, hipStream_t s) : st(s), idx(prof_push(cls, s)) {}
    ~ProfScope() { if (idx >= 0) prof_pop(idx, st); }
};

#ifdef __HIPCC__

struct AssocResult {
    int grid;      // associated SDF or -1
    float err;     // signed distance in metres
    f3 xm;         // point in the model frame
    float w;       // homogeneous weight carried through the transforms
};

// Arg-min over the model's SDFs for one observed point (obsToMod.cu:45-76). The loop index is wave-uniform:
// headers/transforms are scalar loads. Ties keep the lowest SDF index (strict <).
__device__ __forceinline__ AssocResult associate_point(const ModelView &m, float4 xc, float distThr,
                                                       float planeOffset, float pnx, float pny, float pnz) {
    AssocResult r;
    r.grid = -1; r.err = 1e20f; r.w = xc.w;
    const float *__restrict__ T_mc = m.T_mc;
    const float *__restrict__ T_fm = m.T_fm;
    r.xm = se3_apply_w(T_mc, xc.x, xc.y, xc.z, xc.w);
    f3 pn; pn.x = pnx; pn.y = pny; pn.z = pnz;
    if (!(dot(r.xm, pn) >= planeOffset)) return r;
    float best = 1e20f;
    int grid = -1;
    const int nSdfs = m.nSdfs;
    const int slot = keep_slot(distThr);     // SDFs that cannot come within the threshold cannot change the outcome
    for (int g = 0; g < nSdfs; ++g) {
        const int f = m.sdfFrame[g];
        const f3 xf = se3_apply_w(T_fm + 12 * f, r.xm.x, r.xm.y, r.xm.z, xc.w);
        const SdfDev &s = m.sdfs[g];
        const f3 xg = grid_coords(s, xf);
        if (!in_bounds_grad_interp(s, xg) || !in_keep_box(s, slot, xg)) continue;
        const float d = sdf_value<CNT_ASSOC>(m.grids, s, xg) * s.res;
        if (d < best) { best = d; grid = g; }
    }
    if (best * best > distThr * distThr) grid = -1;   // obsToMod.cu:75 (1e20^2 = inf also lands here)
    r.grid = grid; r.err = best;
    return r;
}

// Normal-agreement gate of the single-model association (obsToMod.cu:78-96). Returns true when the point is
// kept; nPred receives the predicted normal in the camera frame.
__device__ __forceinline__ bool normal_test(const ModelView &m, const AssocResult &r, float4 nObs, float normThr,
                                            float4 &nPred) {
    const int f = m.sdfFrame[r.grid];
    const SdfDev &s = m.sdfs[r.grid];
    const float *__restrict__ T_fm = m.T_fm + 12 * f;
    const f3 xf = se3_apply_w(T_fm, r.xm.x, r.xm.y, r.xm.z, r.w);
    const f3 xg = grid_coords(s, xf);
    const f3 gr = sdf_gradient<CNT_ASSOC_GRAD>(m.grids, s, xg);
    float T_mc[12], T_fc[12], T_cf[12], T_f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { T_mc[i] = m.T_mc[i]; T_f[i] = T_fm[i]; }
    se3_mul(T_f, T_mc, T_fc);
    se3_invert(T_fc, T_cf);
    const float invLen = 1.0f / sqrtf(gr.x * gr.x + gr.y * gr.y + gr.z * gr.z + 0.0f * 0.0f);
    const float gx = gr.x * invLen, gy = gr.y * invLen, gz = gr.z * invLen, gw = 0.0f * invLen;
    const f3 n = se3_apply_w(T_cf, gx, gy, gz, gw);
    nPred = make_float4(-1 * n.x, -1 * n.y, -1 * n.z, -1 * gw);
    if (nObs.w > 0.0f) {
        const float dp = nPred.x * nObs.x + nPred.y * nObs.y + nPred.z * nObs.z + nPred.w * 0.0f;
        if (dp < normThr) return false;
    }
    return true;
}

struct Collision {
    bool hit;
    int dstGrid;
    float residual;
    f3 vg;        // grid coordinates in the penetrated SDF
};

// ---- cheap conservative reject of (point, SDF) pairs ------------------------------------------------------------------
// The exact path per pair is a frame transform, three IEEE divisions and a dozen bound tests before the first voxel is
// read, yet nearly every pair is a point of one link against the SDF of a far-away link. Per workgroup and SDF, bounding
// spheres (model frame, {cx, cy, cz, r^2}) of the keep boxes [0..3] and of the interpolation box [4] are put in LDS; a
// point outside the sphere is outside the box for certain (radius grown by far more than the rounding of either path),
// so the pair is dropped with 7 VALU operations and the decisions of the exact path are unchanged.
static constexpr int kSphereSlots = 5;
static constexpr int kSphereStride = 4 * kSphereSlots;     // floats per SDF
__device__ __forceinline__ void make_spheres(const SdfDev &s, const float *T_mf_f, float *out) {
    for (int k = 0; k < kSphereSlots; ++k) {
        float lo[3], hi[3];
        if (k < 4) { for (int a = 0; a < 3; ++a) { lo[a] = s.keepLo[k][a]; hi[a] = s.keepHi[k][a]; } }
        else { lo[0] = lo[1] = lo[2] = s.lo05; hi[0] = s.hi05x; hi[1] = s.hi05y; hi[2] = s.hi05z; }
        float *o = out + 4 * k;
        if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) { o[0] = o[1] = o[2] = 0.0f; o[3] = -1.0f; continue; }   // empty box
        const float cx = s.ox + s.res * 0.5f * (lo[0] + hi[0]), cy = s.oy + s.res * 0.5f * (lo[1] + hi[1]), cz = s.oz + s.res * 0.5f * (lo[2] + hi[2]);
        const f3 c = se3_apply_w(T_mf_f, cx, cy, cz, 1.0f);
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        const float r = 0.5f * s.res * sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f + 1e-4f;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = r * r;
    }
}
__device__ __forceinline__ bool sphere_reject(const float *sph, f3 p) {
    const float dx = p.x - sph[0], dy = p.y - sph[1], dz = p.z - sph[2];
    return dx * dx + dy * dy + dz * dz > sph[3];
}

// First penetrated destination SDF of a point given in the destination model's frame
// (intersection.cu:100-142 / :405-441). pot == nullptr means "test every SDF" (inter-model case); pot may point
// to LDS or global memory. sph (optional): the destination model's spheres (make_spheres, LDS).
__device__ __forceinline__ Collision first_collision(const ModelView &dst, const float *T_fm_dst, f3 v_dm,
                                                     const int *pot, int srcGrid, const float *sph = nullptr) {
    Collision c;
    c.hit = false; c.dstGrid = -1; c.residual = 0.0f; c.vg.x = c.vg.y = c.vg.z = 0.0f;
    const int n = dst.nSdfs;
    if (sph && n <= 64) {
        // Two passes instead of a chain of divergent early-outs per SDF: first a branch-free candidate mask per lane
        // (potential matrix and bounding sphere), whose union over the wave is kept in a scalar; then only the SDFs some
        // lane still wants are visited, in SDF order, so every lane still meets its candidates in the reference's order
        // and stops at its first penetrated one.
        unsigned long long cand = 0ull, any = 0ull;
        for (int g = 0; g < n; ++g) {
            const bool ok = !(pot && !pot[g + srcGrid * n]) && !sphere_reject(sph + kSphereStride * g, v_dm);
            cand |= (unsigned long long)ok << g;
            if (__builtin_amdgcn_ballot_w64(ok) != 0ull) any |= 1ull << g;
        }
        while (any != 0ull) {
            const int g = (int)__builtin_ctzll(any);
            any &= any - 1ull;
            if ((cand >> g) & 1ull) {
                const SdfDev &s = dst.sdfs[g];
                const int f = dst.sdfFrame[g];
                const f3 vd = se3_apply_w(T_fm_dst + 12 * f, v_dm.x, v_dm.y, v_dm.z, 1.0f);
                const f3 vg = grid_coords(s, vd);
                if (in_bounds_grad_interp(s, vg) && in_keep_box(s, 0, vg)) {       // r < 0 needs a negative corner
                    const float r = sdf_value<CNT_INTER>(dst.grids, s, vg) * s.res;
                    if (r < 0) { c.hit = true; c.dstGrid = g; c.residual = r; c.vg = vg; cand = 0ull; }
                }
            }
        }
        return c;
    }
    for (int g = 0; g < n; ++g) {
        if (pot && !pot[g + srcGrid * n]) continue;
        if (sph && sphere_reject(sph + kSphereStride * g, v_dm)) continue;
        const SdfDev &s = dst.sdfs[g];
        const int f = dst.sdfFrame[g];
        const f3 vd = se3_apply_w(T_fm_dst + 12 * f, v_dm.x, v_dm.y, v_dm.z, 1.0f);
        const f3 vg = grid_coords(s, vd);
        if (!in_bounds_grad_interp(s, vg) || !in_keep_box(s, 0, vg)) continue;     // r < 0 needs a negative corner
        const float r = sdf_value<CNT_INTER>(dst.grids, s, vg) * s.res;
        if (r < 0) { c.hit = true; c.dstGrid = g; c.residual = r; c.vg = vg; break; }
    }
    return c;
}

// The same walk when the caller already knows, per lane, which destination SDFs are possible at all (`allowed`: potential matrix and
// a tile-level sphere test folded in) and which some lane of the wave may want (`loop`, wave-uniform): only those are visited, in SDF
// order -- every lane still meets its candidates in the reference's order and stops at its first penetrated one.
__device__ __forceinline__ Collision first_collision_masked(const ModelView &dst, const float *T_fm_dst, f3 v_dm, const float *sph,
                                                            unsigned long long allowed, unsigned long long loop) {
    Collision c;
    c.hit = false; c.dstGrid = -1; c.residual = 0.0f; c.vg.x = c.vg.y = c.vg.z = 0.0f;
    unsigned long long cand = 0ull, any = 0ull;
    for (unsigned long long todo = loop; todo != 0ull; todo &= todo - 1ull) {
        const int g = (int)__builtin_ctzll(todo);
        const bool ok = ((allowed >> g) & 1ull) && !sphere_reject(sph + kSphereStride * g, v_dm);
        cand |= (unsigned long long)ok << g;
        if (__builtin_amdgcn_ballot_w64(ok) != 0ull) any |= 1ull << g;
    }
    while (any != 0ull) {
        const int g = (int)__builtin_ctzll(any);
        any &= any - 1ull;
        if ((cand >> g) & 1ull) {
            const SdfDev &s = dst.sdfs[g];
            const int f = dst.sdfFrame[g];
            const f3 vd = se3_apply_w(T_fm_dst + 12 * f, v_dm.x, v_dm.y, v_dm.z, 1.0f);
            const f3 vg = grid_coords(s, vd);
            if (in_bounds_grad_interp(s, vg) && in_keep_box(s, 0, vg)) {       // r < 0 needs a negative corner
                const float r = sdf_value<CNT_INTER>(dst.grids, s, vg) * s.res;
                if (r < 0) { c.hit = true; c.dstGrid = g; c.residual = r; c.vg = vg; cand = 0ull; }
            }
        }
    }
    return c;
}

// Copy the pose-dependent transforms and the SDF header table into LDS (once per workgroup); used where the
// frame / SDF index differs per lane.
__device__ __forceinline__ void stage_model(const ModelView &m, float *sT_fm, float *sT_mf, SdfDev *sSdf) {
    const int n = 12 * m.nFrames;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { sT_fm[i] = m.T_fm[i]; sT_mf[i] = m.T_mf[i]; }
    if (sSdf) {
        const int nw = m.nSdfs * (int)(sizeof(SdfDev) / 4);
        const int *src = reinterpret_cast<const int *>(m.sdfs);
        int *dst = reinterpret_cast<int *>(sSdf);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    }
}

// doPoseGradientReduction / doParamMapping (+ArticulationOnly), kernel_common.h:73-139.
// off = 6 for full-pose systems, 0 for articulation-only ones.
template <int RED>
__device__ __forceinline__ void reduce_jacobian(float *J, const float *de, int off, int fullDims, int redDims,
                                                const float *__restrict__ A, const int *__restrict__ mapping) {
    const int nf = fullDims - 6, nr = redDims - 6;
    for (int r = 0; r < off; ++r) J[r] = de[r];
    if (RED == 1) {
        for (int r = 0; r < nr; ++r) {
            float acc = 0.0f;
            for (int f = 0; f < nf; ++f) {
                const float mult = A[r + f * nr];
                if (mult == 0.f) continue;
                acc += de[off + f] * mult;
            }
            J[off + r] = acc;
        }
    } else {
        for (int r = 0; r < nr; ++r) J[off + r] = 0.0f;
        for (int f = 0; f < nf; ++f) J[off + mapping[f]] += de[off + f];
    }
}

#endif  // __HIPCC__

}  // namespace dartk
