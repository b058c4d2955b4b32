/* mesh_oracle.c -- CPU restatement of the reference's mesh -> voxel and mesh -> point-cloud steps.
 * TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under dart_amd/). Parity unpinned: the reference holds no
 * test or fixture for either function; this file restates them statement by statement.
 *
 *   orc_splat_solid_mesh   mesh/mesh_splat.cpp:10-265   (splatSolidMesh)
 *   orc_sample_mesh        mesh/mesh_sample.cpp:13-72   (sampleMesh; libc rand(), seeded by the caller: the reference
 *                                                        never calls srand, i.e. runs with the default seed 1)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
static v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 mul3(float s, v3 a) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross3(v3 a, v3 b) { v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
static float min3f(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }   /* std::min(std::min(a,b),c) */
static float max3f(float a, float b, float c) { float m = a < b ? b : a; return m < c ? c : m; }
static double dmax(double a, double b) { return a < b ? b : a; }   /* std::max */
static double dmin(double a, double b) { return b < a ? b : a; }   /* std::min */

/* mesh_splat.cpp:89-111 */
static int crosses(const v3 *verts, const int *face, v3 r0, v3 r1) {
    const float halo = 1e-6f;
    const v3 p1 = verts[face[0]], p2 = verts[face[1]], p3 = verts[face[2]];
    const v3 u = sub3(p2, p1), v = sub3(p3, p1), n = cross3(u, v);
    const float denom = dot3(n, sub3(r1, r0));
    if (denom == 0) return 0;
    const float r = dot3(n, sub3(p1, r0)) / denom;
    if (r < 0 || r > 1) return 0;
    const v3 r_int = add3(r0, mul3(r, sub3(r1, r0)));
    const v3 w = sub3(r_int, p1);
    const float st_denom = dot3(u, v) * dot3(u, v) - dot3(u, u) * dot3(v, v);
    const float s = (dot3(u, v) * dot3(w, v) - dot3(v, v) * dot3(w, u)) / st_denom;
    const float t = (dot3(u, v) * dot3(w, u) - dot3(u, u) * dot3(w, v)) / st_denom;
    if (s < -halo || t < -halo || s + t > (1 + halo)) return 0;
    return 1;
}

/* data[x + y*dx + z*dx*dy] = 0 where at least two of the three parity casts call the voxel filled */
int orc_splat_solid_mesh(const float *vertices, int nVertices, const int *faces, int nFaces, const int *dim,
                         const float *offset, float resolution, float *data) {
    const int dx = dim[0], dy = dim[1], dz = dim[2];
    const size_t vol = (size_t)dx * dy * dz;
    const v3 *verts = (const v3 *)vertices;
    (void)nVertices;
    /* candidate lists (mesh_splat.cpp:13-59) as count + fill passes; faces are appended in face order */
    size_t *start = (size_t *)calloc(vol + 1, sizeof(size_t));
    int *ranges = (int *)malloc((size_t)nFaces * 6 * sizeof(int));
    if (!start || !ranges) return -1;
    for (int f = 0; f < nFaces; ++f) {
        const v3 p1 = verts[faces[3 * f]], p2 = verts[faces[3 * f + 1]], p3 = verts[faces[3 * f + 2]];
        float min_x = min3f(p1.x, p2.x, p3.x), min_y = min3f(p1.y, p2.y, p3.y), min_z = min3f(p1.z, p2.z, p3.z);
        float max_x = max3f(p1.x, p2.x, p3.x), max_y = max3f(p1.y, p2.y, p3.y), max_z = max3f(p1.z, p2.z, p3.z);
        if (min_x == max_x) { if (min_x == offset[0]) max_x += 1e-6; else min_x -= 1e-6; }
        if (min_y == max_y) { if (min_y == offset[1]) max_y += 1e-6; else min_x -= 1e-6; }   /* sic: x */
        if (min_z == max_z) { if (min_z == offset[2]) max_z += 1e-6; else min_z -= 1e-6; }
        int *r = ranges + 6 * f;
        r[0] = (int)dmax(0.0, floor((min_x - offset[0]) / resolution)); r[1] = (int)dmin((double)dx, ceil((max_x - offset[0]) / resolution));
        r[2] = (int)dmax(0.0, floor((min_y - offset[1]) / resolution)); r[3] = (int)dmin((double)dy, ceil((max_y - offset[1]) / resolution));
        r[4] = (int)dmax(0.0, floor((min_z - offset[2]) / resolution)); r[5] = (int)dmin((double)dz, ceil((max_z - offset[2]) / resolution));
        for (int z = r[4]; z < r[5]; ++z)
            for (int y = r[2]; y < r[3]; ++y)
                for (int x = r[0]; x < r[1]; ++x) start[(size_t)x + (size_t)y * dx + (size_t)z * dx * dy + 1]++;
    }
    for (size_t i = 0; i < vol; ++i) start[i + 1] += start[i];
    int *cand = (int *)malloc((start[vol] ? start[vol] : 1) * sizeof(int));
    size_t *fill = (size_t *)malloc(vol * sizeof(size_t));
    char *votes = (char *)calloc(vol, 1);
    char *seen = (char *)malloc(nFaces ? nFaces : 1);
    if (!cand || !fill || !votes || !seen) return -1;
    memcpy(fill, start, vol * sizeof(size_t));
    for (int f = 0; f < nFaces; ++f) {
        const int *r = ranges + 6 * f;
        for (int z = r[4]; z < r[5]; ++z)
            for (int y = r[2]; y < r[3]; ++y)
                for (int x = r[0]; x < r[1]; ++x) cand[fill[(size_t)x + (size_t)y * dx + (size_t)z * dx * dy]++] = f;
    }
    const float res = resolution;
    for (int axis = 0; axis < 3; ++axis) {                 /* x rays, y rays, z rays (mesh_splat.cpp:66-250) */
        const int nA = axis == 0 ? dx : (axis == 1 ? dy : dz);
        const int n1 = axis == 0 ? dy : dx;                 /* inner cross loop */
        const int n2 = axis == 2 ? dy : dz;                 /* outer cross loop */
        for (int o = 0; o < n2; ++o) {
            for (int in = 0; in < n1; ++in) {
                int filling = 0;
                memset(seen, 0, nFaces ? nFaces : 1);
                for (int a = 0; a < nA; ++a) {
                    int x, y, z;
                    if (axis == 0) { x = a; y = in; z = o; }
                    else if (axis == 1) { x = in; y = a; z = o; }
                    else { x = in; y = o; z = a; }
                    int filled = filling;
                    const size_t vox = (size_t)x + (size_t)y * dx + (size_t)z * dx * dy;
                    v3 r0, r1;
                    if (axis == 0) {
                        r0.x = x * res + offset[0];       r0.y = (float)((y + 0.5) * res + offset[1]); r0.z = (float)((z + 0.5) * res + offset[2]);
                        r1.x = (x + 1) * res + offset[0]; r1.y = r0.y;                                 r1.z = r0.z;
                    } else if (axis == 1) {
                        r0.x = (float)((x + 0.5) * res + offset[0]); r0.y = y * res + offset[1];       r0.z = (float)((z + 0.5) * res + offset[2]);
                        r1.x = r0.x;                                 r1.y = (y + 1) * res + offset[1]; r1.z = r0.z;
                    } else {
                        r0.x = (float)((x + 0.5) * res + offset[0]); r0.y = (float)((y + 0.5) * res + offset[1]); r0.z = z * res + offset[2];
                        r1.x = r0.x;                                 r1.y = r0.y;                                 r1.z = (z + 1) * res + offset[2];
                    }
                    for (size_t k = start[vox]; k < start[vox + 1]; ++k) {
                        const int f = cand[k];
                        if (seen[f]) continue;
                        if (!crosses(verts, faces + 3 * f, r0, r1)) continue;
                        seen[f] = 1;
                        filling = !filling;
                        filled = 1;
                    }
                    if (filled) votes[vox]++;
                }
            }
        }
    }
    for (size_t i = 0; i < vol; ++i)
        if (votes[i] >= 2) data[i] = 0;
    free(start); free(ranges); free(cand); free(fill); free(votes); free(seen);
    return 0;
}

/* mesh_sample.cpp:13-72. Returns the number of points written (at most maxPoints; -needed if the buffer is short). */
int orc_sample_mesh(const float *vertices, const int *faces, int nFaces, float sampleDensity, unsigned seed, float *points,
                    int maxPoints) {
    if (nFaces <= 0) return 0;
    const v3 *verts = (const v3 *)vertices;
    double *cum = (double *)malloc((size_t)nFaces * sizeof(double));
    if (!cum) return 0;
    for (int f = 0; f < nFaces; ++f) {
        const v3 A = verts[faces[3 * f]], B = verts[faces[3 * f + 1]], C = verts[faces[3 * f + 2]];
        const v3 ab = sub3(A, B), bc = sub3(B, C), ca = sub3(C, A);
        double a = sqrtf(dot3(ab, ab)), b = sqrtf(dot3(bc, bc)), c = sqrtf(dot3(ca, ca));
        double s = (a + b + c) / 2;
        double area = sqrt(s * (s - a) * (s - b) * (s - c));
        if (isnan(area)) area = 0;
        cum[f] = f == 0 ? area : cum[f - 1] + area;
    }
    const double total = cum[nFaces - 1];
    const int n = (int)round(total * sampleDensity);
    if (n > maxPoints) { free(cum); return -n; }
    srand(seed);
    for (int p = 0; p < n; ++p) {
        double r0 = rand() * total / RAND_MAX;
        int lo = 0, hi = nFaces;                         /* std::lower_bound: first element not less than r0 */
        while (lo < hi) { int mid = lo + (hi - lo) / 2; if (cum[mid] < r0) lo = mid + 1; else hi = mid; }
        const int f = lo;
        const float *A = vertices + 3 * faces[3 * f], *B = vertices + 3 * faces[3 * f + 1], *C = vertices + 3 * faces[3 * f + 2];
        float r1 = rand() / (float)RAND_MAX;
        float r2 = rand() / (float)RAND_MAX;
        for (int i = 0; i < 3; ++i)
            points[3 * p + i] = (1 - sqrtf(r1)) * A[i] + sqrtf(r1) * (1 - r2) * B[i] + sqrtf(r1) * r2 * C[i];
    }
    free(cum);
    return n;
}

/* n draws of libc rand() after srand(seed): pins the product's restatement of glibc's generator */
void orc_libc_rand(unsigned seed, int n, int *out) {
    srand(seed);
    for (int i = 0; i < n; ++i) out[i] = rand();
}
