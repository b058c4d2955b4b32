// dartk_raycast.h -- the ray-cast trace (raycast.cu:121-215, see dartk_mod2obs.hip) as device functions, so that the trace can
// run as a kernel of its own (dartk_mod2obs.hip) or share a launch with the obs->mod term of the single-model loop (dartk_gn.hip).
#pragma once
#include "dartk_internal.h"
#include "dartk_kernels.h"

namespace dartk {

// -----------------------------------------------------------------------------------------------------
struct RayGeom { f3 of, dn; };
#ifdef DARTK_STAMPS
// diagnostic build only: longest march, total steps, marched rays (one copy per translation unit; the stamps build keeps the
// trace in its own kernel in dartk_mod2obs.hip, which is the copy the debug entry point reads)
static __device__ int g_rayMaxSteps;
static __device__ unsigned long long g_raySteps, g_rayMarched;
static __device__ unsigned long long g_rayPhase[6];     // longest wave: passes, cycles of coords / load wait / interpolation+step; all waves: passes, cycles
#endif

// ray of pixel (x,y) expressed in the frame of an SDF whose camera->frame transform is T_fc (raycast.cu:147-161)
__device__ __forceinline__ RayGeom ray_in_frame(const float *T_fc, int x, int y, float flx, float fly, float ppx, float ppy) {
    const float invflx = 1.0f / flx, invfly = 1.0f / fly;
    const float uvx = (float)((double)x + 0.5), uvy = (float)((double)y + 0.5);
    f3 rd; rd.x = (uvx - ppx) * invflx; rd.y = (uvy - ppy) * invfly; rd.z = 1.0f;
    {
        const float inv = 1.0f / sqrtf(rd.x * rd.x + rd.y * rd.y + rd.z * rd.z);
        rd.x *= inv; rd.y *= inv; rd.z *= inv;
    }
    RayGeom g;
    g.of = se3_apply_w(T_fc, 0.f, 0.f, 0.f, 1.f);
    const f3 df = se3_apply_w(T_fc, rd.x, rd.y, rd.z, 0.f);
    const float inv = 1.0f / sqrtf(df.x * df.x + df.y * df.y + df.z * df.z);
    g.dn.x = df.x * inv; g.dn.y = df.y * inv; g.dn.z = df.z * inv;
    return g;
}

// -----------------------------------------------------------------------------------------------------
// Sparse traces. A ray can only find a crossing in an SDF if it passes the SDF's keep box for the level (see below), so per SDF
// only the 8x8 pixel tiles inside the screen rectangle of that box are traced at all; everything outside is "no crossing"
// without a record. Task t of a model = (SDF s, tile k of s's rectangle), numbered SDF by SDF (off[] = exclusive prefix sum of
// the rectangles' tile counts); its 64 records live at trace[t*64 + (y&7)*8 + (x&7)]. The table is recomputed from the pose
// in the prologue of every kernel that needs it (same function, same arithmetic: trace and replay agree by construction):
// no extra launch on the iteration's critical chain, and the reference's dense [nSdfs][pixels] traffic (8 B per pair, written
// and read every iteration, 99 % of it "no crossing") shrinks to the pairs that can matter.
// Layout of the table sR: x0[n] | y0[n] | w[n] | off[n+1]   (tiles).
// -----------------------------------------------------------------------------------------------------
static constexpr int kMaxRaySdfs = 256;
static constexpr int kRectInts = 4 * kMaxRaySdfs + 1;

__device__ __forceinline__ void stage_ray_rects(const ModelView &m, float levelSet, float flx, float fly, float ppx, float ppy,
                                                int width, int height, int *sR) {
    const int n = m.nSdfs;
    const int tilesX = (width + 7) >> 3, tilesY = (height + 7) >> 3;
    const int slot = (levelSet <= 0.0f) ? 0 : keep_slot(levelSet);
    int *sH = sR + 3 * n + 1;        // heights, in the slots the prefix sum overwrites last to first
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const SdfDev &sdf = m.sdfs[s];
        const int f = m.sdfFrame[s];
        float T_cm[12], T_mf[12], T_cf[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { T_cm[i] = m.T_cm[i]; T_mf[i] = m.T_mf[12 * f + i]; }
        se3_mul(T_cm, T_mf, T_cf);
        const float dim[3] = {(float)sdf.dimx, (float)sdf.dimy, (float)sdf.dimz};
        float lo[3], hi[3];
        bool empty = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // the box a marching sample must enter: the keep box where it is interior; where it reaches the planes the samples
            // are clamped to (1 / dim-2) it extends to the face of the grid, beyond which no ray is marched (intersectBox)
            float l = 0.0f, h = dim[a];
            if (slot >= 0) {
                const float kl = sdf.keepLo[slot][a], kh = sdf.keepHi[slot][a];
                if (!(kl <= kh)) empty = true;
                if (kl > 1.0f) l = kl - 0.05f;
                if (kh < dim[a] - 2.0f) h = kh + 0.05f;
            }
            lo[a] = l; hi[a] = h;
        }
        int x0 = 0, y0 = 0, x1 = tilesX - 1, y1 = tilesY - 1;      // tiles, inclusive
        if (empty) { x1 = -1; y1 = -1; }
        else {
            float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
            bool behind = false;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float gx = (c & 1) ? hi[0] : lo[0], gy = (c & 2) ? hi[1] : lo[1], gz = (c & 4) ? hi[2] : lo[2];
                const f3 pc = se3_apply_w(T_cf, sdf.ox + sdf.res * gx, sdf.oy + sdf.res * gy, sdf.oz + sdf.res * gz, 1.0f);
                if (!(pc.z > 1e-3f)) behind = true;
                const float u = pc.x / pc.z * flx + ppx, v = pc.y / pc.z * fly + ppy;
                umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
            if (!behind) {
                // pixel x looks along u = x + 0.5; two pixels of margin cover the rounding of this projection and of the rays
                const float fx0 = floorf(umin - 0.5f) - 2.0f, fx1 = ceilf(umax - 0.5f) + 2.0f;
                const float fy0 = floorf(vmin - 0.5f) - 2.0f, fy1 = ceilf(vmax - 0.5f) + 2.0f;
                const int px0 = (int)fmaxf(fx0, 0.0f), px1 = (int)fminf(fx1, (float)(width - 1));
                const int py0 = (int)fmaxf(fy0, 0.0f), py1 = (int)fminf(fy1, (float)(height - 1));
                if (fx1 < 0.0f || fy1 < 0.0f || fx0 > (float)(width - 1) || fy0 > (float)(height - 1)) { x1 = -1; y1 = -1; }
                else { x0 = px0 >> 3; x1 = px1 >> 3; y0 = py0 >> 3; y1 = py1 >> 3; }
            }
        }
        const int w = x1 >= x0 ? x1 - x0 + 1 : 0, h = y1 >= y0 ? y1 - y0 + 1 : 0;
        sR[s] = x0; sR[n + s] = y0; sR[2 * n + s] = w; sH[s] = h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {         // exclusive prefix sum of the tile counts (a few dozen SDFs)
        int run = 0;
        for (int s = 0; s < n; ++s) {
            const int cnt = sR[2 * n + s] * sH[s];
            sR[3 * n + s] = run;     // overwrites sH[s-1] at most (already consumed)
            run += cnt;
        }
        sR[4 * n] = run;
    }
    __syncthreads();
}

// camera->frame and frame->camera transforms of SDFs [s0, s0+nc) into sT (24 floats each); threads < nc work
__device__ __forceinline__ void replay_stage_transforms(const ModelView &m, int s0, int nc, float *sT) {
    if ((int)threadIdx.x < nc) {
        const int f = m.sdfFrame[s0 + threadIdx.x];
        float T_fm[12], T_mf[12], T_mc[12], T_cm[12], T_fc[12], T_cf[12];
        for (int i = 0; i < 12; ++i) { T_fm[i] = m.T_fm[12 * f + i]; T_mf[i] = m.T_mf[12 * f + i]; T_mc[i] = m.T_mc[i]; }
        se3_invert(T_mc, T_cm);                 // raycast.cu:143
        se3_mul(T_fm, T_mc, T_fc);
        se3_mul(T_cm, T_mf, T_cf);              // T_cm*T_mfs[f]
        for (int i = 0; i < 12; ++i) { sT[24 * threadIdx.x + i] = T_fc[i]; sT[24 * threadIdx.x + 12 + i] = T_cf[i]; }
    }
}

// [per-SDF ray transforms | tile rectangles] of the current pose -> `out` (28 nSdfs + 1 words, the layout ne_mod2obs_kernel's replay
// keeps in LDS), by one whole workgroup; scratch: 28 nSdfs + 1 words of LDS. nSdfs <= blockDim.x.
__device__ __forceinline__ void replay_tables_to_global(const ModelView &m, float flx, float fly, float ppx, float ppy, int width, int height,
                                                        float *scratch, float *__restrict__ out) {
    const int n = m.nSdfs;
    replay_stage_transforms(m, 0, n, scratch);
    stage_ray_rects(m, 0.0f, flx, fly, ppx, ppy, width, height, reinterpret_cast<int *>(scratch + 24 * n));      // (ends with a barrier)
    for (int i = threadIdx.x; i < 28 * n + 1; i += blockDim.x) out[i] = scratch[i];
}

// One ray of one SDF: the set-up (ray in the SDF's frame, intersectBox, keep-box cull) and -- MARCH_NOW -- the march. `s`, `x`, `y`
// may differ per lane (the packed form below marches rays of several tiles and SDFs in one wave); with a wave-uniform `s` the
// tables are read through the scalar cache as before. wantMarch: the ray has to be marched (the set-up alone decides that);
// det / hit: the record ({3e38, 0} = no crossing).
template <bool MARCH_NOW>
__device__ __forceinline__ void trace_one_ray(const ModelView &m, int s, int x, int y, float flx, float fly, float ppx, float ppy, int slot,
                                              float levelSet, bool &wantMarch, float &detOut, float &hitOut) {
    float T_fc[12];
    {
        const int f = m.sdfFrame[s];
        float T_fm[12], T_mc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { T_fm[i] = m.T_fm[12 * f + i]; T_mc[i] = m.T_mc[i]; }
        se3_mul(T_fm, T_mc, T_fc);              // T_fms[f]*T_mc
    }
    const RayGeom g = ray_in_frame(T_fc, x, y, flx, fly, ppx, ppy);
    const f3 of = g.of, dn = g.dn;
    const SdfDev &sdf = m.sdfs[s];
    const float smaxx = sdf.ox + sdf.res * (float)(unsigned)sdf.dimx;
    const float smaxy = sdf.oy + sdf.res * (float)(unsigned)sdf.dimy;
    const float smaxz = sdf.oz + sdf.res * (float)(unsigned)sdf.dimz;
    const float hix = (float)((unsigned)sdf.dimx - 2u), hiy = (float)((unsigned)sdf.dimy - 2u), hiz = (float)((unsigned)sdf.dimz - 2u);
    float t0, t1;   // intersectBox, raycast.cu:11-35
    const float ix = 1.0f / dn.x, iy = 1.0f / dn.y, iz = 1.0f / dn.z;
    {
        const float bx = ix * (sdf.ox - of.x), by = iy * (sdf.oy - of.y), bz = iz * (sdf.oz - of.z);
        const float tx = ix * (smaxx - of.x), ty = iy * (smaxy - of.y), tz = iz * (smaxz - of.z);
        t0 = fmaxf(fmaxf(fminf(tx, bx), fminf(ty, by)), fminf(tz, bz));
        t1 = fminf(fminf(fmaxf(tx, bx), fmaxf(ty, by)), fmaxf(tz, bz));
    }
    float det = 3.0e38f, hit = 0.0f;
    bool march = t1 > t0;
    // A crossing needs a sample with value <= levelSet, i.e. a sample inside the SDF's keep box for that level (SdfDev;
    // samples are clamped to [1, dim-2], so a box side that reaches the clamp planes extends to infinity): a ray that
    // misses the box is not marched at all -- it could only have ended with "no crossing". Most rays of a tile miss the
    // few-centimetre link inside its 10 cm padding. The slabs are tested in frame units with intersectBox's reciprocals (the box
    // carries a margin of 0.05 voxel, five orders of magnitude above the rounding of either form): no division of its own --
    // this test runs for every ray of every tile and is most of what a ray that is not marched costs.
    if (march && slot >= 0) {
        const float o3[3] = {sdf.ox, sdf.oy, sdf.oz}, of3[3] = {of.x, of.y, of.z}, d3[3] = {dn.x, dn.y, dn.z}, i3[3] = {ix, iy, iz};
        const float top[3] = {hix, hiy, hiz};
        float tin = -3.0e38f, tout = 3.0e38f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = sdf.keepLo[slot][a], hi = sdf.keepHi[slot][a];
            if (!(lo <= hi)) { tin = 1.0f; tout = 0.0f; break; }        // nothing at or below the level in this SDF
            const bool openLo = lo <= 1.0f, openHi = hi >= top[a];
            const float Lf = o3[a] + sdf.res * (lo - 0.05f), Hf = o3[a] + sdf.res * (hi + 0.05f);
            float ta = (Lf - of3[a]) * i3[a], tb = (Hf - of3[a]) * i3[a];
            if (openLo) ta = (d3[a] > 0) ? -3.0e38f : 3.0e38f;
            if (openHi) tb = (d3[a] > 0) ? 3.0e38f : -3.0e38f;
            if (d3[a] == 0.0f) {                                            // parallel to the slab: inside or never
                const bool inside = (openLo || of3[a] >= Lf) && (openHi || of3[a] <= Hf);
                ta = inside ? -3.0e38f : 3.0e38f; tb = inside ? 3.0e38f : -3.0e38f;
            }
            tin = fmaxf(tin, fminf(ta, tb));
            tout = fminf(tout, fmaxf(ta, tb));
        }
        march = tin <= tout;
    }
    wantMarch = march;
    if (MARCH_NOW && march) {
        const float *data = m.grids + sdf.dataOff;
        // header fields in registers for the march (the compiler does not hoist loads through the SdfDev reference)
        const float res = sdf.res, rcp = sdf.rcpRes, ox = sdf.ox, oy = sdf.oy, oz = sdf.oz;
        const bool exactDiv = sdf.divExact != 0;
        const int dimx = sdf.dimx, dimy = sdf.dimy, nbx = sdf.nbx, nby = sdf.nby;
        auto eval = [&](float tt) -> float {
            f3 xf, xg;
            xf.x = dn.x * tt + of.x; xf.y = dn.y * tt + of.y; xf.z = dn.z * tt + of.z;
            const float dx = xf.x - ox, dy = xf.y - oy, dz = xf.z - oz;
            if (exactDiv) { xg.x = div_exact(dx, res, rcp); xg.y = div_exact(dy, res, rcp); xg.z = div_exact(dz, res, rcp); }
            else { xg.x = dx / res; xg.y = dy / res; xg.z = dz / res; }
            xg.x = fmaxf(1.0f, fminf(xg.x, hix));
            xg.y = fmaxf(1.0f, fminf(xg.y, hiy));
            xg.z = fmaxf(1.0f, fminf(xg.z, hiz));
            return sdf_lookup<CNT_RAYCAST>(data, dimx, dimy, nbx, nby, xg) * res;
        };
        float sdfValPrev = eval(t0);
        float tPrev = t0;
#ifdef DARTK_STAMPS
        int steps = 0;
#endif
        const float minStep = res / 10;
        // The reference samples the entry point twice (raycast.cu:171-173 before the loop, :180-183 in its first turn): the same point, the
        // same value, a turn that can find no crossing. The second sample is not taken here -- one dependent lookup less per
        // marched ray, a tenth of an average march.
        float sdfVal = sdfValPrev;
        while (t0 < t1 && t0 < 10000.0f) {      // 10000 = the reference's initial closestT
            if (sdfValPrev > levelSet && sdfVal <= levelSet) {
                det = t0;
                hit = (t0 - tPrev) * (levelSet - sdfValPrev) / (sdfVal - sdfValPrev) + tPrev;
                break;      // the reference's loop ends here too: closestT = hit <= t0 < the next t0
            }
            sdfValPrev = sdfVal;
            tPrev = t0;
            t0 += fmaxf(minStep, fabsf(sdfVal));
#ifdef DARTK_STAMPS
            ++steps;
#endif
            if (!(t0 < t1 && t0 < 10000.0f)) break;
            sdfVal = eval(t0);
        }
#ifdef DARTK_STAMPS
        atomicMax(&g_rayMaxSteps, steps);
        atomicAdd(&g_raySteps, (unsigned long long)steps);
        atomicAdd(&g_rayMarched, 1ull);
#endif
    }
    detOut = det; hitOut = hit;
}

// the tasks firstTask, firstTask + taskStride, ... of one wave (sR: the table of stage_ray_rects, already staged and synced)
__device__ __forceinline__ void raycast_trace_tasks(int firstTask, int taskStride, const int *sR, float flx, float fly, float ppx, float ppy,
                                                    int width, int height, const ModelView &m, float2 *__restrict__ trace, float levelSet) {
    const int n = m.nSdfs;
    const int nTasks = sR[4 * n];
    const int lane = threadIdx.x & 63;
    const int slot = (levelSet <= 0.0f) ? 0 : keep_slot(levelSet);
    int s = 0;                                       // SDF of the current task: tasks are visited in increasing order
    for (int task = __builtin_amdgcn_readfirstlane(firstTask); task < nTasks; task += taskStride) {
        while (task >= sR[3 * n + s + 1]) ++s;       // wave-uniform
        const int local = task - sR[3 * n + s], rw = sR[2 * n + s];
        const int x = (sR[s] + local % rw) * 8 + (lane & 7);
        const int y = (sR[n + s] + local / rw) * 8 + (lane >> 3);
        if (x >= width || y >= height) continue;
        bool want; float det, hit;
        trace_one_ray<true>(m, s, x, y, flx, fly, ppx, ppy, slot, levelSet, want, det, hit);
        count_lookup<CNT_OTHER>();                  // (diagnostic build: trace records written)
        trace[(size_t)task * 64 + lane] = make_float2(det, hit);
    }
}

}  // namespace dartk
