// dartk_mesh.hip -- solid-mesh voxel splat on the device (the model/asset pipeline's only grid-sized step besides
// the distance transform). Replaces: mesh/mesh_splat.cpp:10-265 (splatSolidMesh: per-voxel candidate face lists,
// parity ray casts along x, y and z, a voxel is inside when at least two of the three casts say so).
//
// The reference walks every ray voxel by voxel over per-voxel candidate lists and keeps a set of the faces the ray
// has already crossed. A face is a candidate on one contiguous voxel interval of a ray (its bounding-box range), so
// "first voxel of the interval whose segment test passes" is the voxel where that face toggles the ray's parity;
// the order of the faces inside a voxel does not matter for the parity nor for the "filled" flag. One thread owns
// one ray: it marks the toggle voxels in its own row of a scratch grid, then sweeps the row. No atomics.
// Arithmetic: fp32 in the reference's operation order (the library is built without contraction); the voxel-centre
// coordinates are formed in double and narrowed, as `(y+0.5)*resolution + offset` is there.
#include "dartk_internal.h"

namespace dartk {

__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }

struct FaceRec {            // per face, computed once: voxel ranges [lo,hi) of the candidate box and the plane basis
    int lo[3], hi[3];
    f3 p1, u, v, n;
    float uv, uu, vv, stDenom;
};

__device__ inline int range_lo(float mn, float off, float res) {
    const double q = floor((double)((mn - off) / res));
    const double b = 0.0 < q ? q : 0.0;                     // std::max(0.0, q); NaN -> 0.0
    return b < 2.0e9 ? (int)b : 2000000000;                 // beyond any grid: the loop is empty either way
}
__device__ inline int range_hi(float mx, float off, float res, int dim) {
    const double q = ceil((double)((mx - off) / res));
    const double b = q < (double)dim ? q : (double)dim;     // std::min((double)dim, q); NaN -> dim
    return b > -2.0e9 ? (int)b : -2000000000;               // the loop runs while (double)i < b, b integral and <= dim
}

__global__ void mesh_face_setup_kernel(const float *__restrict__ verts, const int *__restrict__ faces, int nFaces, int dx,
                                       int dy, int dz, float ox, float oy, float oz, float res, int nVertices,
                                       FaceRec *__restrict__ rec, int *__restrict__ badFaces) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nFaces) return;
    const int i1 = faces[3 * f], i2 = faces[3 * f + 1], i3 = faces[3 * f + 2];
    if (i1 < 0 || i2 < 0 || i3 < 0 || i1 >= nVertices || i2 >= nVertices || i3 >= nVertices) {   // never dereferenced
        FaceRec e = {};
        rec[f] = e;         // empty ranges: no ray considers it
        atomicAdd(badFaces, 1);
        return;
    }
    const f3 p1 = {verts[3 * i1], verts[3 * i1 + 1], verts[3 * i1 + 2]};
    const f3 p2 = {verts[3 * i2], verts[3 * i2 + 1], verts[3 * i2 + 2]};
    const f3 p3 = {verts[3 * i3], verts[3 * i3 + 1], verts[3 * i3 + 2]};
    float min_x = fminf(fminf(p1.x, p2.x), p3.x), max_x = fmaxf(fmaxf(p1.x, p2.x), p3.x);
    float min_y = fminf(fminf(p1.y, p2.y), p3.y), max_y = fmaxf(fmaxf(p1.y, p2.y), p3.y);
    float min_z = fminf(fminf(p1.z, p2.z), p3.z), max_z = fmaxf(fmaxf(p1.z, p2.z), p3.z);
    // axis-aligned faces get a sliver of thickness (mesh_splat.cpp:29-49; the y case widens x, as written there)
    if (min_x == max_x) { if (min_x == ox) max_x = (float)((double)max_x + 1e-6); else min_x = (float)((double)min_x - 1e-6); }
    if (min_y == max_y) { if (min_y == oy) max_y = (float)((double)max_y + 1e-6); else min_x = (float)((double)min_x - 1e-6); }
    if (min_z == max_z) { if (min_z == oz) max_z = (float)((double)max_z + 1e-6); else min_z = (float)((double)min_z - 1e-6); }
    FaceRec r;
    r.lo[0] = range_lo(min_x, ox, res); r.hi[0] = range_hi(max_x, ox, res, dx);
    r.lo[1] = range_lo(min_y, oy, res); r.hi[1] = range_hi(max_y, oy, res, dy);
    r.lo[2] = range_lo(min_z, oz, res); r.hi[2] = range_hi(max_z, oz, res, dz);
    r.p1 = p1;
    r.u = p2 - p1;
    r.v = p3 - p1;
    r.n = cross(r.u, r.v);
    r.uv = dot(r.u, r.v); r.uu = dot(r.u, r.u); r.vv = dot(r.v, r.v);
    r.stDenom = r.uv * r.uv - r.uu * r.vv;
    rec[f] = r;
}

// the segment test of one voxel (mesh_splat.cpp:89-111): true when the face is crossed inside [r0, r1]
__device__ inline bool segment_hits_face(const FaceRec &fr, const f3 r0, const f3 r1) {
    const float halo = 1e-6f;
    const f3 dir = r1 - r0;
    const float denom = dot(fr.n, dir);
    if (denom == 0) return false;
    const float r = dot(fr.n, fr.p1 - r0) / denom;
    if (r < 0 || r > 1) return false;
    const f3 rInt = r0 + r * dir;
    const f3 w = rInt - fr.p1;
    const float wv = dot(w, fr.v), wu = dot(w, fr.u);
    const float s = (fr.uv * wv - fr.vv * wu) / fr.stDenom;
    const float t = (fr.uv * wu - fr.uu * wv) / fr.stDenom;
    if (s < -halo || t < -halo || s + t > (1 + halo)) return false;
    return true;
}

// axis 0: rays along x, one per (y,z); axis 1: along y per (x,z); axis 2: along z per (x,y). toggles / filled are
// [3][dz][dy][dx] bytes; a thread only touches the voxels of its own ray.
__global__ void mesh_ray_parity_kernel(const FaceRec *__restrict__ rec, int nFaces, int dx, int dy, int dz, float ox, float oy,
                                       float oz, float res, unsigned char *__restrict__ toggles,
                                       unsigned char *__restrict__ filled) {
    const int axis = blockIdx.y;
    const int dim[3] = {dx, dy, dz};
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;          // the two cross axes
    const int nRays = dim[a1] * dim[a2];
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= nRays) return;
    int c[3];
    c[a1] = ray % dim[a1];
    c[a2] = ray / dim[a1];
    const float off[3] = {ox, oy, oz};
    const size_t vol = (size_t)dx * dy * dz;
    unsigned char *tg = toggles + axis * vol, *fl = filled + axis * vol;
    const size_t stride[3] = {1, (size_t)dx, (size_t)dx * dy};
    const size_t base = c[a1] * stride[a1] + c[a2] * stride[a2];
    const int len = dim[axis];
    // voxel-centre coordinates of the two cross axes: double arithmetic, narrowed (mesh_splat.cpp:79-80)
    float mid[3];
    mid[a1] = (float)(((double)c[a1] + 0.5) * (double)res + (double)off[a1]);
    mid[a2] = (float)(((double)c[a2] + 0.5) * (double)res + (double)off[a2]);
    for (int f = 0; f < nFaces; ++f) {
        const FaceRec fr = rec[f];
        if (c[a1] < fr.lo[a1] || c[a1] >= fr.hi[a1] || c[a2] < fr.lo[a2] || c[a2] >= fr.hi[a2]) continue;
        for (int i = fr.lo[axis]; i < fr.hi[axis]; ++i) {
            float p0[3], p1[3];
            p0[a1] = p1[a1] = mid[a1];
            p0[a2] = p1[a2] = mid[a2];
            p0[axis] = (float)i * res + off[axis];
            p1[axis] = (float)(i + 1) * res + off[axis];
            const f3 r0 = {p0[0], p0[1], p0[2]}, r1 = {p1[0], p1[1], p1[2]};
            if (segment_hits_face(fr, r0, r1)) {        // bit 0: parity of the faces first crossed here, bit 1: any
                unsigned char &t = tg[base + i * stride[axis]];
                t = (unsigned char)(((t ^ 1) & 1) | 2);
                break;
            }
        }
    }
    bool filling = false;
    for (int i = 0; i < len; ++i) {
        const unsigned char n = tg[base + i * stride[axis]];
        fl[base + i * stride[axis]] = (filling || (n & 2)) ? 1 : 0;
        if (n & 1) filling = !filling;
    }
}

__global__ void mesh_vote_kernel(const unsigned char *__restrict__ filled, size_t vol, float *__restrict__ sdf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vol) return;
    if (filled[i] + filled[vol + i] + filled[2 * vol + i] >= 2) sdf[i] = 0;
}

}  // namespace dartk

using namespace dartk;

extern "C" {

int dartk_splat_solid_mesh(const float *d_vertices, int nVertices, const int32_t *d_faces, int nFaces, int dimX, int dimY,
                           int dimZ, const float *h_offset, float resolution, float *d_sdf, void *stream) {
    DARTK_REQUIRE(d_vertices && d_faces && h_offset && d_sdf, "dartk_splat_solid_mesh: null argument");
    DARTK_REQUIRE(nVertices > 0 && nFaces >= 0 && dimX > 0 && dimY > 0 && dimZ > 0 && resolution > 0,
                  "dartk_splat_solid_mesh: bad size (%d vertices, %d faces, grid %d x %d x %d, resolution %g)", nVertices,
                  nFaces, dimX, dimY, dimZ, (double)resolution);
    DARTK_REQUIRE(nFaces < 128 * 1024 * 1024, "dartk_splat_solid_mesh: too many faces");
    if (nFaces == 0) return DARTK_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t vol = (size_t)dimX * dimY * dimZ;
    FaceRec *rec = nullptr;
    unsigned char *bytes = nullptr;
    int bad = 0;
    DARTK_HIP(hipMalloc((void **)&rec, (size_t)nFaces * sizeof(FaceRec)));
    if (hipMalloc((void **)&bytes, 6 * vol + 16) != hipSuccess) {
        (void)hipFree(rec);
        DARTK_REQUIRE(false, "dartk_splat_solid_mesh: out of device memory (%zu bytes of scratch)", 6 * vol);
    }
    unsigned char *toggles = bytes, *filled = bytes + 3 * vol;
    int *d_bad = reinterpret_cast<int *>(bytes + ((6 * vol + 3) & ~(size_t)3));
    hipError_t e = hipMemsetAsync(toggles, 0, 3 * vol, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mesh_face_setup_kernel, dim3((nFaces + 127) / 128), dim3(128), 0, st, d_vertices, d_faces, nFaces,
                           dimX, dimY, dimZ, h_offset[0], h_offset[1], h_offset[2], resolution, nVertices, rec, d_bad);
        const int d01 = dimX * dimY, d12 = dimY * dimZ, d02 = dimX * dimZ;
        const int most = d01 > d12 ? (d01 > d02 ? d01 : d02) : (d12 > d02 ? d12 : d02);
        hipLaunchKernelGGL(mesh_ray_parity_kernel, dim3((most + 63) / 64, 3), dim3(64), 0, st, rec, nFaces, dimX, dimY, dimZ,
                           h_offset[0], h_offset[1], h_offset[2], resolution, toggles, filled);
        hipLaunchKernelGGL(mesh_vote_kernel, dim3((unsigned)((vol + 255) / 256)), dim3(256), 0, st, filled, vol, d_sdf);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);      // the scratch is released below
    }
    (void)hipFree(rec);
    (void)hipFree(bytes);
    DARTK_HIP(e);
    DARTK_REQUIRE(bad == 0, "dartk_splat_solid_mesh: %d face(s) index vertices outside [0,%d)", bad, nVertices);
    return DARTK_OK;
}

}  // extern "C"
