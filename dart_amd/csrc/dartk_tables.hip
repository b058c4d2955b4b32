// dartk_tables.hip -- the boundary for reference call-sites that pass RAW DEVICE POINTER TABLES instead of a MirroredModel:
// errorAndDataAssociationMultiModel (obsToMod.h:88-107, called from optimizer.cpp:620-630 with the tracker's tables of
// tracker.cpp:238-243), raycastPrediction (raycast.h:9-22, called from prediction_renderer.cpp:195-207) and
// countSelfIntersections (intersection.h:11-18). A MirroredModel backed by this library hands out the library's own device
// arrays through the getters below (mirrored_model.h:65-87); the *_tables entry points take exactly the reference's argument
// lists, find the model handle that owns the arrays, and forward to the handle-based entry points. No reference call-site has
// to change: only MirroredModel's getters and the three free functions are re-implemented.
#include "dartk_internal.h"
#include "dartk_kernels.h"

#include <mutex>
#include <vector>

namespace dartk {

static std::mutex g_regMutex;
static std::vector<Model *> g_models;

void registry_add(Model *m) { std::lock_guard<std::mutex> l(g_regMutex); g_models.push_back(m); }
void registry_remove(Model *m) {
    std::lock_guard<std::mutex> l(g_regMutex);
    for (size_t i = 0; i < g_models.size(); ++i)
        if (g_models[i] == m) { g_models.erase(g_models.begin() + i); return; }
}
static Model *registry_find(const void *p) {
    if (!p) return nullptr;
    std::lock_guard<std::mutex> l(g_regMutex);
    for (Model *m : g_models) {
        const ModelView &v = m->v;
        if (p == v.T_fm || p == v.T_mf || p == v.sdfFrame || p == v.sdfs || p == v.T_mc || p == v.obsData) return m;
    }
    return nullptr;
}

}  // namespace dartk

using namespace dartk;

extern "C" {

const float *dartk_model_device_T_mf(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.T_mf : nullptr; }
const float *dartk_model_device_T_fm(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.T_fm : nullptr; }
const float *dartk_model_device_T_mc(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.T_mc : nullptr; }
const int32_t *dartk_model_device_sdf_frames(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.sdfFrame : nullptr; }
const void *dartk_model_device_sdfs(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.sdfs : nullptr; }
const int32_t *dartk_model_device_frame_parents(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.frameParent : nullptr; }
const int32_t *dartk_model_device_joint_types(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.jointType : nullptr; }
const float *dartk_model_device_joint_axes(dartk_model_t h) { return h ? reinterpret_cast<Model *>(h)->v.jointAxis : nullptr; }

dartk_model_t dartk_model_from_device_pointer(const void *d_ptr) { return reinterpret_cast<dartk_model_t>(registry_find(d_ptr)); }

int dartk_assoc_multi_tables(const float *d_vert, const float *d_norm, int width, int height, int nModels, const float *d_T_mcs,
                             const void *const *d_T_fms, const void *const *d_sdfFrames, const void *const *d_sdfs,
                             const int32_t *d_nSdfs, const float *d_distanceThresholds, const float *d_normalThresholds,
                             const float *d_planeOffsets, const float *d_planeNormals3, int32_t *d_lastElements,
                             dartk_da_point *const *d_pts, int32_t *d_dbgDA, float *d_dbgErr, float *d_dbgNorm, void *stream) {
    (void)d_sdfFrames; (void)d_sdfs; (void)d_nSdfs; (void)d_normalThresholds; (void)d_dbgNorm;      // never read by the reference kernel either (SURVEY A-20)
    DARTK_REQUIRE(d_vert && d_T_fms && d_distanceThresholds && d_planeOffsets && d_planeNormals3 && d_lastElements && d_pts,
                  "dartk_assoc_multi_tables: null argument");
    DARTK_REQUIRE(nModels >= 1 && nModels <= DARTK_MAX_MODELS, "nModels %d out of range [1,%d]", nModels, DARTK_MAX_MODELS);
    hipStream_t st = as_stream(stream);
    const void *tab[DARTK_MAX_MODELS];
    dartk_da_point *pts[DARTK_MAX_MODELS];
    float thr[DARTK_MAX_MODELS], po[DARTK_MAX_MODELS], pn[3 * DARTK_MAX_MODELS];
    // the tables live in device memory (MirroredVector::devicePtr()): a few small blocking copies, as everywhere in the reference
    DARTK_HIP(hipMemcpyAsync(tab, d_T_fms, nModels * sizeof(void *), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipMemcpyAsync(pts, d_pts, nModels * sizeof(void *), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipMemcpyAsync(thr, d_distanceThresholds, nModels * sizeof(float), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipMemcpyAsync(po, d_planeOffsets, nModels * sizeof(float), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipMemcpyAsync(pn, d_planeNormals3, 3 * nModels * sizeof(float), hipMemcpyDeviceToHost, st));
    DARTK_HIP(hipStreamSynchronize(st));
    dartk_model_t handles[DARTK_MAX_MODELS];
    for (int m = 0; m < nModels; ++m) {
        Model *mod = registry_find(tab[m]);
        DARTK_REQUIRE(mod, "dartk_assoc_multi_tables: T_fms[%d] is not a device array of a live dartk model", m);
        handles[m] = reinterpret_cast<dartk_model_t>(mod);
    }
    // T_mcs is authoritative for this launch, as in the reference (optimizer.cpp:603-608 uploads the models' T_mc as a kernel
    // argument), and only for it: the models' own T_mc / T_cm stay as they are
    return gn_assoc_multi_entry(d_vert, d_norm, width, height, nModels, handles, d_T_mcs, thr, po, pn, pts, d_lastElements, d_dbgDA, d_dbgErr, stream);
}

int dartk_raycast_prediction_tables(float flx, float fly, float ppx, float ppy, int width, int height, int modelNum,
                                    const float *h_T_mc, const void *d_T_fms, const void *d_T_mfs, const void *d_sdfFrames,
                                    const void *d_sdfs, int nSdfs, float *d_prediction, float levelSet, void *stream) {
    (void)d_T_mfs; (void)d_sdfFrames; (void)d_sdfs;
    Model *mod = registry_find(d_T_fms);
    DARTK_REQUIRE(mod, "dartk_raycast_prediction_tables: T_fms is not a device array of a live dartk model");
    DARTK_REQUIRE(nSdfs == mod->v.nSdfs, "dartk_raycast_prediction_tables: nSdfs %d does not match the model's %d", nSdfs, mod->v.nSdfs);
    if (h_T_mc) {           // SE3 passed by value in the reference: made the model's camera->model transform (and its inverse)
        hipStream_t st = as_stream(stream);
        const float *A = h_T_mc;
        float T[24];
        for (int i = 0; i < 12; ++i) T[i] = A[i];
        T[12] = A[0]; T[13] = A[4]; T[14] = A[8];  T[15] = -(A[0] * A[3] + A[4] * A[7] + A[8] * A[11]);
        T[16] = A[1]; T[17] = A[5]; T[18] = A[9];  T[19] = -(A[1] * A[3] + A[5] * A[7] + A[9] * A[11]);
        T[20] = A[2]; T[21] = A[6]; T[22] = A[10]; T[23] = -(A[2] * A[3] + A[6] * A[7] + A[10] * A[11]);
        DARTK_HIP(hipMemcpyAsync(mod->v.T_mc, T, 12 * sizeof(float), hipMemcpyHostToDevice, st));
        DARTK_HIP(hipMemcpyAsync(mod->v.T_cm, T + 12, 12 * sizeof(float), hipMemcpyHostToDevice, st));
        DARTK_HIP(hipStreamSynchronize(st));        // T is a stack temporary
    }
    return dartk_raycast_prediction(flx, fly, ppx, ppy, width, height, modelNum, reinterpret_cast<dartk_model_t>(mod), d_prediction, levelSet, stream);
}

int dartk_count_self_intersections_tables(const float *d_sites, int nSites, const void *d_T_mfs, const void *d_T_fms,
                                          const void *d_sdfFrames, const void *d_sdfs, int nSdfs,
                                          const int32_t *d_potentialIntersection, int32_t *h_count) {
    (void)d_T_fms; (void)d_sdfFrames; (void)d_sdfs;
    Model *mod = registry_find(d_T_mfs);
    DARTK_REQUIRE(mod, "dartk_count_self_intersections_tables: T_mfs is not a device array of a live dartk model");
    DARTK_REQUIRE(nSdfs == mod->v.nSdfs, "dartk_count_self_intersections_tables: nSdfs %d does not match the model's %d", nSdfs, mod->v.nSdfs);
    return dartk_count_self_intersections(d_sites, nSites, reinterpret_cast<dartk_model_t>(mod), d_potentialIntersection, h_count, nullptr);
}

}  // extern "C"
