// extern "C" boundary of libcpt_amd.so (declarations and contracts: include/cpt_amd.h).
#include "../../include/cpt_amd.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "host/film.hpp"
#include "host/loader.hpp"
#include "host/multi_gpu.hpp"
#include "host/pathtracer.hpp"
#include "host/scene.hpp"
#include "lbvh.h"
#include "render.h"

namespace {
thread_local std::string g_error;

int Fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
int HipFail(hipError_t e, const char *what) { return Fail(CPT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }

template <typename F> int Guard(F &&fn) {
    try {
        return fn();
    } catch (const std::invalid_argument &e) {
        return Fail(CPT_ERR_INVALID, e.what());
    } catch (const std::exception &e) {
        return Fail(CPT_ERR_DEVICE, e.what());
    } catch (...) {
        return Fail(CPT_ERR_DEVICE, "unknown failure");
    }
}

cpt::Bbox MakeBox(const float b[8]) {
    cpt::Bbox r;
    memcpy(&r, b, 32);
    return r;
}
}  // namespace

struct cpt_scene {
    Scene scene;
};

struct cpt_tracer {
    cpt_scene *owner;
    std::unique_ptr<Film> film;
    std::unique_ptr<PathTracer> tracer;
};

struct cpt_comm {
    std::unique_ptr<cpt::ShardComm> comm;
};

struct cpt_multi {
    cpt_scene *owner;
    std::unique_ptr<Film> film;
    std::unique_ptr<MultiPathTracer> tracer;
};

struct cpt_builder {
    cpt::BuildWorkspace ws;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

extern "C" {

const char *cpt_last_error(void) { return g_error.c_str(); }

int cpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cpt_set_device(int device) {
    hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipSetDevice");
}

void *cpt_malloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { HipFail(e, "hipMalloc"); return nullptr; }
    return p;
}
int cpt_free(void *d_ptr) {
    hipError_t e = hipFree(d_ptr);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipFree");
}
int cpt_memcpy_h2d(void *d_dst, const void *src, size_t bytes) {
    // complete on return, also as seen from non-blocking streams (kernels of a tracer run on one)
    hipError_t e = hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipMemcpy H2D");
}
int cpt_memcpy_d2h(void *dst, const void *d_src, size_t bytes) {
    hipError_t e = hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipMemcpy D2H");
}
int cpt_memset(void *d_dst, int value, size_t bytes) {
    hipError_t e = hipMemset(d_dst, value, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipMemset");
}
int cpt_synchronize(void) {
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipDeviceSynchronize");
}
int cpt_memory_info(size_t *free_bytes, size_t *total_bytes) {
    if (!free_bytes || !total_bytes) return Fail(CPT_ERR_INVALID, "cpt_memory_info: null argument");
    hipError_t e = hipMemGetInfo(free_bytes, total_bytes);
    return e == hipSuccess ? CPT_OK : HipFail(e, "hipMemGetInfo");
}

// ------------------------------------------------------------------ LBVH
int cpt_build_accel(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives) {
    return cpt_build_accel_ex(d_nodes, d_bboxes, merged_bbox, num_primitives, nullptr, nullptr, nullptr);
}

int cpt_build_accel_ex(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives, uint64_t *d_keys_out,
                       uint32_t *d_parents_out, float *stage_ms) {
    if (num_primitives == 0) return CPT_OK;
    if (!d_nodes || !d_bboxes || !merged_bbox) return Fail(CPT_ERR_INVALID, "cpt_build_accel: null argument");
    if (num_primitives > 0x7fffffffu) return Fail(CPT_ERR_INVALID, "cpt_build_accel: more than 2^31-1 primitives");
    int rc = cpt::lbvh_build(static_cast<cpt::AccelNode *>(d_nodes), static_cast<cpt::Bbox *>(d_bboxes), MakeBox(merged_bbox), num_primitives, nullptr,
                             nullptr, nullptr, d_keys_out, d_parents_out, stage_ms);
    if (rc != 0) return HipFail((hipError_t)rc, "lbvh_build");
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? CPT_OK : HipFail(e, "lbvh_build");
}

int cpt_build_accel_noalloc(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives) {
    if (num_primitives == 0) return CPT_OK;
    if (!d_nodes || !d_bboxes || !merged_bbox) return Fail(CPT_ERR_INVALID, "cpt_build_accel_noalloc: null argument");
    if (num_primitives > cpt::lbvh_static_max_primitives())
        return Fail(CPT_ERR_INVALID, "cpt_build_accel_noalloc: " + std::to_string(num_primitives) + " primitives, the fixed scratch array holds at most " +
                                         std::to_string(cpt::lbvh_static_max_primitives()) + " (use cpt_build_accel)");
    int rc = cpt::lbvh_build_static(static_cast<cpt::AccelNode *>(d_nodes), static_cast<cpt::Bbox *>(d_bboxes), MakeBox(merged_bbox), num_primitives, nullptr);
    return rc == 0 ? CPT_OK : HipFail((hipError_t)rc, "lbvh_build_static");
}
uint32_t cpt_build_accel_noalloc_max_primitives(void) { return cpt::lbvh_static_max_primitives(); }

cpt_builder *cpt_builder_create(void) { return new cpt_builder(); }

void cpt_builder_destroy(cpt_builder *b) {
    if (!b) return;
    b->ws.release();
    if (b->e0) (void)hipEventDestroy(b->e0);
    if (b->e1) (void)hipEventDestroy(b->e1);
    delete b;
}

int cpt_builder_set_sort(cpt_builder *b, int variant) {
    if (!b || variant < 0 || variant > 1) return Fail(CPT_ERR_INVALID, "cpt_builder_set_sort: variant must be 0 or 1");
    b->ws.sort_variant = variant;
    return CPT_OK;
}

int cpt_builder_set_carrier_limit(cpt_builder *b, uint32_t limit) {
    if (!b) return Fail(CPT_ERR_INVALID, "cpt_builder_set_carrier_limit: null builder");
    b->ws.carrier_limit = limit;
    return CPT_OK;
}

int cpt_builder_build(cpt_builder *b, void *d_nodes, void *d_bboxes, const void *d_leaf_boxes, const float merged_bbox[8],
                      uint32_t num_primitives, float *elapsed_ms, float *stage_ms) {
    if (!b || !d_nodes || !d_bboxes || !merged_bbox) return Fail(CPT_ERR_INVALID, "cpt_builder_build: null argument");
    if (num_primitives == 0) return CPT_OK;
    if (num_primitives > 0x7fffffffu) return Fail(CPT_ERR_INVALID, "cpt_builder_build: more than 2^31-1 primitives");
    if (elapsed_ms) {
        if (!b->e0) { hipError_t e = hipEventCreate(&b->e0); if (e != hipSuccess) return HipFail(e, "hipEventCreate"); }
        if (!b->e1) { hipError_t e = hipEventCreate(&b->e1); if (e != hipSuccess) return HipFail(e, "hipEventCreate"); }
        int rc0 = b->ws.reserve(num_primitives);     // keep the allocation out of the timed region
        if (rc0) return HipFail((hipError_t)rc0, "workspace");
        (void)hipEventRecord(b->e0, nullptr);
    }
    int rc = cpt::lbvh_build(static_cast<cpt::AccelNode *>(d_nodes), static_cast<cpt::Bbox *>(d_bboxes), MakeBox(merged_bbox), num_primitives, nullptr,
                             &b->ws, static_cast<const cpt::Bbox *>(d_leaf_boxes), nullptr, nullptr, stage_ms);
    if (rc != 0) return HipFail((hipError_t)rc, "lbvh_build");
    if (elapsed_ms) {
        (void)hipEventRecord(b->e1, nullptr);
        hipError_t e = hipEventSynchronize(b->e1);
        if (e != hipSuccess) return HipFail(e, "hipEventSynchronize");
        (void)hipEventElapsedTime(elapsed_ms, b->e0, b->e1);
    }
    return CPT_OK;
}

int cpt_builder_build_records(cpt_builder *b, void *d_nodes, void *d_bboxes, const void *d_leaf_boxes, const float merged_bbox[8], uint32_t num_primitives,
                              int top_level, void *d_binary_records, void *d_two_level_records, void *d_root_record) {
    if (!b || !d_nodes || !d_bboxes || !merged_bbox || !d_root_record) return Fail(CPT_ERR_INVALID, "cpt_builder_build_records: null argument");
    if (!d_binary_records && !d_two_level_records) return Fail(CPT_ERR_INVALID, "cpt_builder_build_records: neither binary nor two-level records asked for");
    if (num_primitives == 0) return CPT_OK;
    if (num_primitives > 0x7fffffffu) return Fail(CPT_ERR_INVALID, "cpt_builder_build_records: more than 2^31-1 primitives");
    cpt::RecordOutputs records;
    records.binary = static_cast<cpt::PackedNode *>(d_binary_records);
    records.wide = static_cast<cpt::PackedNode *>(d_two_level_records);
    records.root = static_cast<cpt::AccelRoot *>(d_root_record);
    records.top_level = top_level ? 1 : 0;
    int rc = cpt::lbvh_build(static_cast<cpt::AccelNode *>(d_nodes), static_cast<cpt::Bbox *>(d_bboxes), MakeBox(merged_bbox), num_primitives, nullptr,
                             &b->ws, static_cast<const cpt::Bbox *>(d_leaf_boxes), nullptr, nullptr, nullptr, &records);
    return rc == 0 ? CPT_OK : HipFail((hipError_t)rc, "lbvh_build");
}

// ------------------------------------------------------------------ scene
cpt_scene *cpt_scene_create(void) { return new cpt_scene(); }
void cpt_scene_destroy(cpt_scene *s) { delete s; }

int cpt_scene_load_obj(cpt_scene *s, const char *path, int bsdf_type) {
    if (!s || !path) return Fail(CPT_ERR_INVALID, "cpt_scene_load_obj: null argument");
    return Guard([&] { return LoadObjScene(s->scene, path, bsdf_type) ? CPT_OK : Fail(CPT_ERR_IO, std::string("failed to load obj file '") + path + "'"); });
}
int cpt_scene_load_obj_cached(cpt_scene *s, const char *path, int bsdf_type, int *from_cache) {
    if (!s || !path) return Fail(CPT_ERR_INVALID, "cpt_scene_load_obj_cached: null argument");
    return Guard([&] {
        bool hit = false;
        if (!LoadObjSceneCached(s->scene, path, bsdf_type, &hit)) return Fail(CPT_ERR_IO, std::string("failed to load obj file '") + path + "'");
        if (from_cache) *from_cache = hit ? 1 : 0;
        return (int)CPT_OK;
    });
}
int cpt_scene_load_json(cpt_scene *s, const char *path) {
    if (!s || !path) return Fail(CPT_ERR_INVALID, "cpt_scene_load_json: null argument");
    return Guard([&] { return LoadJsonScene(s->scene, path) ? CPT_OK : Fail(CPT_ERR_IO, std::string("failed to load json file '") + path + "'"); });
}
int cpt_scene_load_xml(cpt_scene *s, const char *path) {
    if (!s || !path) return Fail(CPT_ERR_INVALID, "cpt_scene_load_xml: null argument");
    return Guard([&] { return LoadXmlScene(s->scene, path) ? CPT_OK : Fail(CPT_ERR_IO, std::string("failed to load xml file '") + path + "'"); });
}

int cpt_scene_add_object(cpt_scene *s, const char *name, const float *positions, const float *normals, const float *texcoords,
                         uint32_t num_vertices, const uint32_t *indices, uint32_t num_triangles, const float material16[16],
                         const float *transform16) {
    if (!s || !positions || !normals || !texcoords || !indices || !material16) return Fail(CPT_ERR_INVALID, "cpt_scene_add_object: null argument");
    for (size_t i = 0; i < 3 * (size_t)num_triangles; i++)
        if (indices[i] >= num_vertices) return Fail(CPT_ERR_INVALID, "cpt_scene_add_object: vertex index out of range");
    if (transform16) {
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (memcmp(transform16, ident, sizeof(ident)) != 0)
            return Fail(CPT_ERR_INVALID, "cpt_scene_add_object: only identity transforms can be given here; set translate/rotate/scale on the object");
    }
    return Guard([&] {
        auto object = s->scene.AddObject(name ? name : "");
        auto mesh = std::make_shared<Mesh>();
        std::vector<cpt::V3> pos(num_vertices), nrm(num_vertices);
        std::vector<cpt::V2> uv(num_vertices);
        memcpy(pos.data(), positions, 12 * (size_t)num_vertices);
        memcpy(nrm.data(), normals, 12 * (size_t)num_vertices);
        memcpy(uv.data(), texcoords, 8 * (size_t)num_vertices);
        mesh->SetPositions(std::move(pos));
        mesh->SetNormals(std::move(nrm));
        mesh->SetTexcoords(std::move(uv));
        mesh->SetIndices(std::vector<uint32_t>(indices, indices + 3 * (size_t)num_triangles));
        object->AddComponent<MeshComponent>()->SetMesh(mesh);
        auto mat = std::make_shared<Material>();
        mat->name = name ? name : "";
        mat->emission = cpt::V3{material16[0], material16[1], material16[2]};
        mat->diffuse = cpt::V3{material16[3], material16[4], material16[5]};
        mat->specular = cpt::V3{material16[6], material16[7], material16[8]};
        mat->transmittance = cpt::V3{material16[9], material16[10], material16[11]};
        mat->shininess = material16[12];
        mat->ior = material16[13];
        mat->opacity = material16[14];
        mat->bsdf_type = (int)material16[15];
        mat->SetChanged();
        object->AddComponent<MaterialComponent>()->SetMaterial(mat);
        return CPT_OK;
    });
}

int cpt_scene_set_camera(cpt_scene *s, const float pos[3], const float look_at[3], const float up[3], float fov_degrees, uint32_t film_width,
                         uint32_t film_height) {
    if (!s || !pos || !look_at || !up) return Fail(CPT_ERR_INVALID, "cpt_scene_set_camera: null argument");
    auto existing = s->scene.FirstObjectWith<CameraComponent>();
    auto object = existing ? existing : s->scene.AddObject("camera");
    CameraComponent *c = existing ? object->GetComponent<CameraComponent>() : object->AddComponent<CameraComponent>();
    c->pos = cpt::V3{pos[0], pos[1], pos[2]};
    c->look_at = cpt::V3{look_at[0], look_at[1], look_at[2]};
    c->up = cpt::V3{up[0], up[1], up[2]};
    c->fov = fov_degrees;
    c->film_width = film_width;
    c->film_height = film_height;
    return CPT_OK;
}

static SceneObject *MeshObject(const cpt_scene *s, uint32_t index) {
    uint32_t k = 0;
    for (auto &obj : s->scene.Objects())
        if (obj->HasComponent<MeshComponent>() && obj->HasComponent<MaterialComponent>()) {
            if (k == index) return obj.get();
            k++;
        }
    return nullptr;
}

uint32_t cpt_scene_num_objects(const cpt_scene *s) {
    if (!s) return 0;
    uint32_t k = 0;
    for (auto &obj : s->scene.Objects()) k += (obj->HasComponent<MeshComponent>() && obj->HasComponent<MaterialComponent>()) ? 1 : 0;
    return k;
}

int cpt_scene_object_info(const cpt_scene *s, uint32_t index, char *name_buf, size_t name_buf_size, uint32_t *num_vertices, uint32_t *num_triangles,
                          float material16[16], float transform16[16]) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj) return Fail(CPT_ERR_INVALID, "cpt_scene_object_info: no such object");
    const Mesh *mesh = obj->GetComponent<MeshComponent>()->GetMesh();
    const Material *mat = obj->GetComponent<MaterialComponent>()->GetMaterial();
    if (name_buf && name_buf_size) {
        strncpy(name_buf, obj->Name().c_str(), name_buf_size - 1);
        name_buf[name_buf_size - 1] = '\0';
    }
    if (num_vertices) *num_vertices = (uint32_t)mesh->VericesCount();
    if (num_triangles) *num_triangles = (uint32_t)(mesh->IndicesCount() / 3);
    if (material16) {
        const float m[16] = {mat->emission.x, mat->emission.y, mat->emission.z, mat->diffuse.x, mat->diffuse.y, mat->diffuse.z,
                             mat->specular.x, mat->specular.y, mat->specular.z, mat->transmittance.x, mat->transmittance.y, mat->transmittance.z,
                             mat->shininess, mat->ior, mat->opacity, (float)mat->bsdf_type};
        memcpy(material16, m, sizeof(m));
    }
    if (transform16) {
        cpt::M4 t = obj->GetTransform();
        memcpy(transform16, &t, 64);
    }
    return CPT_OK;
}

int cpt_scene_object_set_trs(cpt_scene *s, uint32_t index, const float translate[3], const float rotate_degrees[3], const float scale[3]) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj) return Fail(CPT_ERR_INVALID, "cpt_scene_object_set_trs: no such object");
    if (!translate || !rotate_degrees || !scale) return Fail(CPT_ERR_INVALID, "cpt_scene_object_set_trs: null argument");
    obj->translate = cpt::V3{translate[0], translate[1], translate[2]};
    obj->rotate = cpt::V3{rotate_degrees[0], rotate_degrees[1], rotate_degrees[2]};
    obj->scale = cpt::V3{scale[0], scale[1], scale[2]};
    return CPT_OK;
}

int cpt_scene_object_set_material(cpt_scene *s, uint32_t index, const float material16[16]) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj || !material16) return Fail(CPT_ERR_INVALID, "cpt_scene_object_set_material: no such object or null argument");
    Material *mat = obj->GetComponent<MaterialComponent>()->GetMaterial();
    mat->emission = cpt::V3{material16[0], material16[1], material16[2]};
    mat->diffuse = cpt::V3{material16[3], material16[4], material16[5]};
    mat->specular = cpt::V3{material16[6], material16[7], material16[8]};
    mat->transmittance = cpt::V3{material16[9], material16[10], material16[11]};
    mat->shininess = material16[12];
    mat->ior = material16[13];
    mat->opacity = material16[14];
    mat->bsdf_type = (int)material16[15];
    mat->SetChanged();
    return CPT_OK;
}

int cpt_scene_object_transforms(const cpt_scene *s, uint32_t index, float transform16[16], float inverse16[16]) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj) return Fail(CPT_ERR_INVALID, "cpt_scene_object_transforms: no such object");
    const cpt::M4 t = obj->GetTransform();
    if (transform16) memcpy(transform16, &t, 64);
    if (inverse16) {
        const cpt::M4 inv = InverseTransform(t);
        memcpy(inverse16, &inv, 64);
    }
    return CPT_OK;
}

int cpt_scene_object_copy(const cpt_scene *s, uint32_t index, float *positions, float *normals, float *texcoords, uint32_t *indices) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj) return Fail(CPT_ERR_INVALID, "cpt_scene_object_copy: no such object");
    const Mesh *mesh = obj->GetComponent<MeshComponent>()->GetMesh();
    if (positions) memcpy(positions, mesh->Positions().data(), 12 * mesh->Positions().size());
    if (normals) memcpy(normals, mesh->Normals().data(), 12 * mesh->Normals().size());
    if (texcoords) memcpy(texcoords, mesh->Texcoords().data(), 8 * mesh->Texcoords().size());
    if (indices) memcpy(indices, mesh->Indices().data(), 4 * mesh->Indices().size());
    return CPT_OK;
}

int cpt_scene_camera(const cpt_scene *s, float cam10[10], uint32_t res2[2]) {
    if (!s) return Fail(CPT_ERR_INVALID, "cpt_scene_camera: null scene");
    auto obj = s->scene.FirstObjectWith<CameraComponent>();
    if (!obj) return Fail(CPT_ERR_INVALID, "cpt_scene_camera: the scene has no camera");
    const CameraComponent *c = obj->GetComponent<CameraComponent>();
    if (cam10) {
        const float v[10] = {c->pos.x, c->pos.y, c->pos.z, c->look_at.x, c->look_at.y, c->look_at.z, c->up.x, c->up.y, c->up.z, c->fov};
        memcpy(cam10, v, sizeof(v));
    }
    if (res2) { res2[0] = c->film_width; res2[1] = c->film_height; }
    return CPT_OK;
}

// ------------------------------------------------------------------ tracer
cpt_tracer *cpt_tracer_create(cpt_scene *s, uint32_t film_width, uint32_t film_height) {
    if (!s || film_width == 0 || film_height == 0) { Fail(CPT_ERR_INVALID, "cpt_tracer_create: bad argument"); return nullptr; }
    auto *t = new cpt_tracer();
    t->owner = s;
    t->film = std::make_unique<Film>(film_width, film_height);
    t->tracer = std::make_unique<PathTracer>(s->scene, *t->film);
    return t;
}
void cpt_tracer_destroy(cpt_tracer *t) { delete t; }

#define TRACER_OR_FAIL(t) do { if (!(t)) return Fail(CPT_ERR_INVALID, "null tracer"); } while (0)

int cpt_tracer_set_max_depth(cpt_tracer *t, int max_depth) { TRACER_OR_FAIL(t); t->tracer->SetMaxDepth(max_depth); return CPT_OK; }
int cpt_tracer_set_channel(cpt_tracer *t, int channel) { TRACER_OR_FAIL(t); t->tracer->SetChannel(channel); return CPT_OK; }
int cpt_tracer_set_integrator(cpt_tracer *t, int mode) {
    TRACER_OR_FAIL(t);
    if (mode < 0 || mode > 5) return Fail(CPT_ERR_INVALID, "cpt_tracer_set_integrator: mode must be 0 to 5");
    t->tracer->SetIntegrator(mode);
    return CPT_OK;
}
int cpt_tracer_set_traversal(cpt_tracer *t, int mode) {
    TRACER_OR_FAIL(t);
    if (mode < 0 || mode > 1) return Fail(CPT_ERR_INVALID, "cpt_tracer_set_traversal: 0 (exact) or 1 (fast)");
    t->tracer->SetTraversal(mode);
    return CPT_OK;
}
int cpt_tracer_set_node_records(cpt_tracer *t, int mode) {
    TRACER_OR_FAIL(t);
    if (mode < 0 || mode > 1) return Fail(CPT_ERR_INVALID, "cpt_tracer_set_node_records: 0 (binary records only) or 1 (two-level records too)");
    t->tracer->SetNodeRecords(mode);
    return CPT_OK;
}
int cpt_tracer_set_wavefront_tail(cpt_tracer *t, int bounces) { TRACER_OR_FAIL(t); t->tracer->SetWavefrontTail(bounces); return CPT_OK; }
int cpt_tracer_set_profile_build(cpt_tracer *t, int on) { TRACER_OR_FAIL(t); t->tracer->SetProfileBuild(on != 0); return CPT_OK; }
int cpt_tracer_build_buffers(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->BuildBuffers(); return CPT_OK; });
}
int cpt_tracer_update(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->Update(); return CPT_OK; });
}
int cpt_tracer_reset_accumulation(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->ResetAccumelation(); t->film->Clear(t->tracer->Stream()); t->tracer->NoteFilmCleared(); return CPT_OK; });
}
int cpt_tracer_synchronize(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->Synchronize(); return CPT_OK; });
}
uint32_t cpt_tracer_current_spp(const cpt_tracer *t) { return t ? t->tracer->CurrentSpp() : 0; }
uint32_t cpt_tracer_num_lights(const cpt_tracer *t) { return t ? t->tracer->NumLights() : 0; }

int cpt_tracer_render_samples(cpt_tracer *t, uint32_t spp_begin, uint32_t spp_count, int sum_mode, float *d_output, uint32_t width, uint32_t height) {
    TRACER_OR_FAIL(t);
    if (spp_begin == 0) return Fail(CPT_ERR_INVALID, "cpt_tracer_render_samples: sample indices are 1-based");
    if (d_output && (width == 0 || height == 0)) return Fail(CPT_ERR_INVALID, "cpt_tracer_render_samples: zero-sized output");
    return Guard([&] { t->tracer->RenderSamples(spp_begin, spp_count, sum_mode != 0, d_output, width, height); return CPT_OK; });
}

int cpt_path_tracer_render(const cpt_render_params *params) {
    if (!params || !params->scene) return Fail(CPT_ERR_INVALID, "cpt_path_tracer_render: null argument");
    if (!params->output || params->screen_width == 0 || params->screen_height == 0 || params->spp == 0 || params->channel < 0 || params->channel > 1)
        return Fail(CPT_ERR_INVALID, "cpt_path_tracer_render: bad parameter block");
    return Guard([&] {
        params->scene->tracer->RenderOneSample(params->output, params->screen_width, params->screen_height, params->spp, params->max_depth, params->channel);
        return CPT_OK;
    });
}

float *cpt_tracer_film_device_ptr(cpt_tracer *t) {
    if (!t) return nullptr;
    t->tracer->NoteFilmExposed();
    try { return t->film->DeviceMap(t->tracer->Stream()); } catch (const std::exception &e) { Fail(CPT_ERR_DEVICE, e.what()); return nullptr; }
}
int cpt_tracer_film_read(cpt_tracer *t, float *rgba) {
    TRACER_OR_FAIL(t);
    if (!rgba) return Fail(CPT_ERR_INVALID, "cpt_tracer_film_read: null buffer");
    return Guard([&] { t->tracer->Synchronize(); t->film->ReadBack(rgba); return CPT_OK; });
}
int cpt_tracer_film_save(cpt_tracer *t, const char *path) {
    TRACER_OR_FAIL(t);
    if (!path) return Fail(CPT_ERR_INVALID, "cpt_tracer_film_save: null path");
    return Guard([&] { t->tracer->Synchronize(); return t->film->SaveTo(path) ? CPT_OK : Fail(CPT_ERR_IO, std::string("cannot write '") + path + "'"); });
}

int cpt_tracer_first_hit(cpt_tracer *t, const float *d_rays, uint64_t num_rays, uint32_t *d_instance, uint32_t *d_primitive, float *d_t, float *d_uv) {
    TRACER_OR_FAIL(t);
    if (num_rays == 0) return CPT_OK;
    if (!d_rays || !d_instance || !d_primitive) return Fail(CPT_ERR_INVALID, "cpt_tracer_first_hit: null argument");
    if (t->tracer->NumInstances() == 0) return Fail(CPT_ERR_INVALID, "cpt_tracer_first_hit: build_buffers has not produced any geometry");
    return Guard([&] { t->tracer->FirstHit(d_rays, num_rays, d_instance, d_primitive, d_t, d_uv); return CPT_OK; });
}
int cpt_tracer_set_query_kernel(cpt_tracer *t, int mode) {
    TRACER_OR_FAIL(t);
    if (mode < 0 || mode > 1) return Fail(CPT_ERR_INVALID, "cpt_tracer_set_query_kernel: 0 (one ray per lane for the launch) or 1 (persistent, lanes refilled)");
    t->tracer->SetQueryKernel(mode);
    return CPT_OK;
}
int cpt_tracer_set_scene_in_lds(cpt_tracer *t, int on) {
    TRACER_OR_FAIL(t);
    t->tracer->SetSceneInLds(on != 0);
    return t->tracer->SceneInLds() ? 1 : 0;
}
int cpt_tracer_occlude(cpt_tracer *t, const float *d_rays, uint64_t num_rays, uint8_t *d_occluded) {
    TRACER_OR_FAIL(t);
    if (num_rays == 0) return CPT_OK;
    if (!d_rays || !d_occluded) return Fail(CPT_ERR_INVALID, "cpt_tracer_occlude: null argument");
    if (t->tracer->NumInstances() == 0) return Fail(CPT_ERR_INVALID, "cpt_tracer_occlude: build_buffers has not produced any geometry");
    return Guard([&] {
        int rc = cpt::launch_occlude(t->tracer->DeviceSceneForQueries(), d_rays, num_rays, d_occluded, t->tracer->Stream());
        return rc == 0 ? CPT_OK : HipFail((hipError_t)rc, "occlude");
    });
}
int cpt_tracer_camera_rays(cpt_tracer *t, uint32_t width, uint32_t height, uint32_t spp, float *d_rays) {
    TRACER_OR_FAIL(t);
    if (!d_rays || spp == 0) return Fail(CPT_ERR_INVALID, "cpt_tracer_camera_rays: bad argument");
    int rc = cpt::launch_camera_rays(t->tracer->DeviceScene().camera, width, height, spp, d_rays, t->tracer->Stream());
    return rc == 0 ? CPT_OK : HipFail((hipError_t)rc, "camera_rays");
}

int64_t cpt_tracer_get_accel(cpt_tracer *t, int mesh, uint32_t *nodes, float *bboxes, uint64_t *keys) {
    if (!t) return Fail(CPT_ERR_INVALID, "null tracer");
    PathTracer::AccelView v = t->tracer->Accel(mesh);
    if (!v.nodes) return Fail(CPT_ERR_INVALID, "cpt_tracer_get_accel: no such structure");
    int rc = Guard([&] {
        t->tracer->Synchronize();
        size_t nn = 2 * (size_t)v.num_primitives - 1;
        if (nodes) CPT_HIP(hipMemcpy(nodes, v.nodes, 8 * nn, hipMemcpyDeviceToHost));
        if (bboxes) CPT_HIP(hipMemcpy(bboxes, v.bboxes, 32 * nn, hipMemcpyDeviceToHost));
        if (keys) CPT_HIP(hipMemcpy(keys, v.keys, 8 * (size_t)v.num_primitives, hipMemcpyDeviceToHost));
        return CPT_OK;
    });
    return rc == CPT_OK ? (int64_t)v.num_primitives : rc;
}

int cpt_tracer_update_many(cpt_tracer *t, uint32_t count) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->UpdateMany(count); return CPT_OK; });
}
int cpt_tracer_set_samples_per_launch(cpt_tracer *t, uint32_t samples) {
    TRACER_OR_FAIL(t);
    if (samples < 1 || samples > 4096) return Fail(CPT_ERR_INVALID, "cpt_tracer_set_samples_per_launch: 1 to 4096");
    t->tracer->SetSamplesPerLaunch(samples);
    return CPT_OK;
}
int cpt_tracer_update_timed(cpt_tracer *t, uint32_t count, float *elapsed_ms) {
    TRACER_OR_FAIL(t);
    if (!elapsed_ms) return Fail(CPT_ERR_INVALID, "cpt_tracer_update_timed: null argument");
    return Guard([&] { *elapsed_ms = t->tracer->UpdateTimed(count); return CPT_OK; });
}

int cpt_tracer_timer_begin(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->TimerBegin(); return CPT_OK; });
}
int cpt_tracer_timer_end(cpt_tracer *t, float *elapsed_ms) {
    TRACER_OR_FAIL(t);
    if (!elapsed_ms) return Fail(CPT_ERR_INVALID, "cpt_tracer_timer_end: null argument");
    return Guard([&] { *elapsed_ms = t->tracer->TimerEnd(); return CPT_OK; });
}

int cpt_tracer_counters(cpt_tracer *t, uint64_t out5[5]) {
    TRACER_OR_FAIL(t);
    return Guard([&] {
        unsigned long long c[6];
        t->tracer->ReadCounters(c);
        for (int i = 0; i < 5; i++) out5[i] = c[i];
        return CPT_OK;
    });
}
int cpt_tracer_stack_overflows(cpt_tracer *t, uint64_t *count) {
    TRACER_OR_FAIL(t);
    if (!count) return Fail(CPT_ERR_INVALID, "cpt_tracer_stack_overflows: null argument");
    return Guard([&] {
        unsigned long long c[6];
        t->tracer->ReadCounters(c);
        *count = c[5];
        return CPT_OK;
    });
}
int cpt_tracer_reset_counters(cpt_tracer *t) {
    TRACER_OR_FAIL(t);
    return Guard([&] { t->tracer->ResetCounters(); return CPT_OK; });
}
int cpt_tracer_build_stage_ms(cpt_tracer *t, float out5[5]) {
    TRACER_OR_FAIL(t);
    memcpy(out5, t->tracer->BuildStageMs(), 5 * sizeof(float));
    return CPT_OK;
}
int cpt_tracer_build_extra_ms(cpt_tracer *t, float out2[2]) {
    TRACER_OR_FAIL(t);
    if (!out2) return Fail(CPT_ERR_INVALID, "cpt_tracer_build_extra_ms: null argument");
    out2[0] = t->tracer->LeafBoxesMs();
    out2[1] = t->tracer->PackMs();
    return CPT_OK;
}

// ------------------------------------------------------------------ several GPUs
int cpt_shard_samples(uint32_t total, uint32_t rank, uint32_t world, uint32_t first, uint32_t *begin, uint32_t *count) {
    if (!begin || !count || world == 0 || rank >= world || first == 0) return Fail(CPT_ERR_INVALID, "cpt_shard_samples: bad argument");
    const cpt::SampleBlock b = cpt::split_samples(total, rank, world, first);
    *begin = b.begin;
    *count = b.count;
    return CPT_OK;
}

int cpt_comm_unique_id(uint8_t id[128]) {
    if (!id) return Fail(CPT_ERR_INVALID, "cpt_comm_unique_id: null argument");
    return Guard([&] {
        auto v = cpt::ShardComm::UniqueId();
        memcpy(id, v.data(), v.size());
        return CPT_OK;
    });
}

cpt_comm *cpt_comm_create(const uint8_t id[128], int rank, int world) {
    if (!id) { Fail(CPT_ERR_INVALID, "cpt_comm_create: null id"); return nullptr; }
    cpt_comm *c = nullptr;
    int rc = Guard([&] {
        std::array<char, cpt::kCommIdBytes> v;
        memcpy(v.data(), id, v.size());
        auto comm = cpt::ShardComm::InitRank(v, rank, world);
        c = new cpt_comm{std::move(comm)};
        return CPT_OK;
    });
    return rc == CPT_OK ? c : nullptr;
}

void cpt_comm_destroy(cpt_comm *c) { delete c; }

int cpt_comm_census(const cpt_comm *c, int *ranks, int *rank, int *device) {
    if (!c) return Fail(CPT_ERR_INVALID, "cpt_comm_census: null communicator");
    return Guard([&] {
        const cpt::ShardComm::CensusResult seen = c->comm->Census();
        if (ranks) *ranks = seen.ranks;
        if (rank) *rank = seen.rank;
        if (device) *device = seen.device;
        return CPT_OK;
    });
}

int cpt_tracer_reduce_film(cpt_tracer *t, cpt_comm *c, int root, int finish) {
    TRACER_OR_FAIL(t);
    if (!c || root < 0 || root >= c->comm->World()) return Fail(CPT_ERR_INVALID, "cpt_tracer_reduce_film: bad communicator or root");
    return Guard([&] { cpt::ReduceFilmToRoot(*t->tracer, *t->film, *c->comm, root, finish != 0); return CPT_OK; });
}

cpt_multi *cpt_multi_create(cpt_scene *s, uint32_t film_width, uint32_t film_height, const int *shard_devices, uint32_t num_shards) {
    if (!s || film_width == 0 || film_height == 0 || !shard_devices || num_shards == 0) { Fail(CPT_ERR_INVALID, "cpt_multi_create: bad argument"); return nullptr; }
    cpt_multi *m = nullptr;
    int rc = Guard([&] {
        auto holder = std::make_unique<cpt_multi>();
        holder->owner = s;
        holder->film = std::make_unique<Film>(film_width, film_height);
        holder->tracer = std::make_unique<MultiPathTracer>(s->scene, *holder->film, std::vector<int>(shard_devices, shard_devices + num_shards));
        m = holder.release();
        return CPT_OK;
    });
    return rc == CPT_OK ? m : nullptr;
}

void cpt_multi_destroy(cpt_multi *m) {
    if (!m) return;
    m->tracer.reset();       // before the film its first shard renders into
    delete m;
}

#define MULTI_OR_FAIL(m) do { if (!(m)) return Fail(CPT_ERR_INVALID, "null multi-GPU tracer"); } while (0)

int cpt_multi_set_max_depth(cpt_multi *m, int max_depth) { MULTI_OR_FAIL(m); m->tracer->SetMaxDepth(max_depth); return CPT_OK; }
int cpt_multi_set_integrator(cpt_multi *m, int mode) {
    MULTI_OR_FAIL(m);
    if (mode < 0 || mode > 5) return Fail(CPT_ERR_INVALID, "cpt_multi_set_integrator: mode must be 0 to 5");
    m->tracer->SetIntegrator(mode);
    return CPT_OK;
}
int cpt_multi_set_traversal(cpt_multi *m, int mode) {
    MULTI_OR_FAIL(m);
    if (mode < 0 || mode > 1) return Fail(CPT_ERR_INVALID, "cpt_multi_set_traversal: 0 (exact) or 1 (fast)");
    m->tracer->SetTraversal(mode);
    return CPT_OK;
}
int cpt_multi_set_node_records(cpt_multi *m, int mode) {
    MULTI_OR_FAIL(m);
    if (mode < 0 || mode > 1) return Fail(CPT_ERR_INVALID, "cpt_multi_set_node_records: 0 (binary records only) or 1 (two-level records too)");
    m->tracer->SetNodeRecords(mode);
    return CPT_OK;
}
int cpt_multi_set_samples_per_launch(cpt_multi *m, uint32_t samples) {
    MULTI_OR_FAIL(m);
    if (samples < 1 || samples > 4096) return Fail(CPT_ERR_INVALID, "cpt_multi_set_samples_per_launch: 1 to 4096");
    m->tracer->SetSamplesPerLaunch(samples);
    return CPT_OK;
}
int cpt_multi_build_buffers(cpt_multi *m) {
    MULTI_OR_FAIL(m);
    return Guard([&] { m->tracer->BuildBuffers(); return CPT_OK; });
}
int cpt_multi_render(cpt_multi *m, uint32_t total_spp, float *elapsed_ms) {
    MULTI_OR_FAIL(m);
    return Guard([&] {
        m->tracer->Render(total_spp);
        if (elapsed_ms) *elapsed_ms = m->tracer->LastRenderMs();
        return CPT_OK;
    });
}
int cpt_multi_film_read(cpt_multi *m, float *rgba) {
    MULTI_OR_FAIL(m);
    if (!rgba) return Fail(CPT_ERR_INVALID, "cpt_multi_film_read: null buffer");
    return Guard([&] { m->tracer->Synchronize(); m->film->ReadBack(rgba); return CPT_OK; });
}
int cpt_multi_film_save(cpt_multi *m, const char *path) {
    MULTI_OR_FAIL(m);
    if (!path) return Fail(CPT_ERR_INVALID, "cpt_multi_film_save: null path");
    return Guard([&] { m->tracer->Synchronize(); return m->film->SaveTo(path) ? CPT_OK : Fail(CPT_ERR_IO, std::string("cannot write '") + path + "'"); });
}
int cpt_multi_counters(cpt_multi *m, uint64_t out6[6]) {
    MULTI_OR_FAIL(m);
    if (!out6) return Fail(CPT_ERR_INVALID, "cpt_multi_counters: null argument");
    return Guard([&] {
        unsigned long long c[6];
        m->tracer->ReadCounters(c);
        for (int i = 0; i < 6; i++) out6[i] = c[i];
        return CPT_OK;
    });
}
int cpt_multi_reset_counters(cpt_multi *m) {
    MULTI_OR_FAIL(m);
    return Guard([&] { m->tracer->ResetCounters(); return CPT_OK; });
}
uint32_t cpt_multi_num_devices(const cpt_multi *m) { return m ? (uint32_t)m->tracer->NumDevices() : 0; }

int cpt_selftest_division(uint64_t pairs, uint32_t seed, uint64_t *mismatches) {
    if (!mismatches) return Fail(CPT_ERR_INVALID, "cpt_selftest_division: null argument");
    return Guard([&] {
        cpt::HipBuffer counter(sizeof(unsigned long long));
        counter.Zero();
        int rc = cpt::selftest_division(pairs, seed, counter.TypedGpuData<unsigned long long>(), nullptr);
        if (rc != 0) return HipFail((hipError_t)rc, "selftest_division");
        unsigned long long host = 0;
        CPT_HIP(hipMemcpy(&host, counter.GpuData(), sizeof(host), hipMemcpyDeviceToHost));
        *mismatches = host;
        return CPT_OK;
    });
}

int cpt_stream_copy_rate(uint64_t bytes, int repeats, double *gb_per_s) {
    if (!gb_per_s || repeats < 1 || bytes < 16) return Fail(CPT_ERR_INVALID, "cpt_stream_copy_rate: bad argument");
    return Guard([&] {
        cpt::HipBuffer src(bytes), dst(bytes);
        src.Zero();
        hipEvent_t e0, e1;
        CPT_HIP(hipEventCreate(&e0));
        CPT_HIP(hipEventCreate(&e1));
        int rc = cpt::launch_stream_copy(src.GpuData(), dst.GpuData(), bytes, nullptr);       // warm-up
        if (rc == 0) {
            CPT_HIP(hipEventRecord(e0, nullptr));
            for (int k = 0; k < repeats && rc == 0; k++) rc = cpt::launch_stream_copy(src.GpuData(), dst.GpuData(), bytes, nullptr);
            CPT_HIP(hipEventRecord(e1, nullptr));
            CPT_HIP(hipEventSynchronize(e1));
        }
        float ms = 0.0f;
        if (rc == 0) CPT_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (rc != 0) return HipFail((hipError_t)rc, "stream_copy");
        *gb_per_s = 2.0 * (double)(bytes / 16 * 16) * repeats / ((double)ms * 1e-3) / 1e9;     // bytes read + bytes written
        return (int)CPT_OK;
    });
}

int cpt_valu_issue_rate(int mix, int waves_per_simd, double *ginst_per_s, double *cycles_per_inst, double *shader_mhz) {
    if (!ginst_per_s || mix < 0 || mix > 2 || waves_per_simd < 1 || waves_per_simd > 8) return Fail(CPT_ERR_INVALID, "cpt_valu_issue_rate: mix 0, 1 or 2, 1 to 8 waves per SIMD");
    return Guard([&] {
        int device = 0, cus = 0;
        CPT_HIP(hipGetDevice(&device));
        CPT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        const uint32_t iterations = 4096;                      // x 64 instructions per trip
        const size_t waves = (size_t)cus * waves_per_simd * 4;
        cpt::HipBuffer sink(sizeof(float)), cycles(sizeof(unsigned long long) * waves);
        cycles.Zero();
        hipEvent_t e0, e1;
        CPT_HIP(hipEventCreate(&e0));
        CPT_HIP(hipEventCreate(&e1));
        int rc = cpt::launch_valu_rate(mix, waves_per_simd, cus, iterations, sink.TypedGpuData<float>(), cycles.TypedGpuData<unsigned long long>(), nullptr);   // warm-up: clocks up
        const int repeats = 5;
        if (rc == 0) {
            CPT_HIP(hipEventRecord(e0, nullptr));
            for (int k = 0; k < repeats && rc == 0; k++)
                rc = cpt::launch_valu_rate(mix, waves_per_simd, cus, iterations, sink.TypedGpuData<float>(), cycles.TypedGpuData<unsigned long long>(), nullptr);
            CPT_HIP(hipEventRecord(e1, nullptr));
            CPT_HIP(hipEventSynchronize(e1));
        }
        float ms = 0.0f;
        if (rc == 0) CPT_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (rc != 0) return HipFail((hipError_t)rc, "valu_rate");
        std::vector<unsigned long long> host(waves);
        CPT_HIP(hipMemcpy(host.data(), cycles.GpuData(), sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
        double sum = 0.0;
        for (unsigned long long c : host) sum += (double)c;
        const double per_wave = (double)iterations * 64.0;
        const double seconds = (double)ms * 1e-3 / repeats;
        *ginst_per_s = per_wave * (double)waves / seconds / 1e9;
        if (cycles_per_inst) *cycles_per_inst = sum / (double)waves / per_wave;
        // a launch lasts as long as its waves do (they all start together): shader clock = a wave's cycles / the launch time
        if (shader_mhz) *shader_mhz = sum / (double)waves / seconds / 1e6;
        return (int)CPT_OK;
    });
}

int cpt_image_load(const char *path, uint32_t *width, uint32_t *height, uint32_t *rgba, size_t capacity) {
    if (!path || !width || !height) return Fail(CPT_ERR_INVALID, "cpt_image_load: null argument");
    return Guard([&] {
        std::string error;
        auto image = LoadImage(path, error);
        if (!image) return Fail(CPT_ERR_IO, error);
        *width = image->width;
        *height = image->height;
        if (rgba) {
            if (capacity < image->rgba.size()) return Fail(CPT_ERR_INVALID, "cpt_image_load: buffer too small");
            memcpy(rgba, image->rgba.data(), image->rgba.size() * sizeof(uint32_t));
        }
        return (int)CPT_OK;
    });
}

int cpt_image_save_png(const char *path, const uint32_t *rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || width == 0 || height == 0) return Fail(CPT_ERR_INVALID, "cpt_image_save_png: bad argument");
    return Guard([&] {
        Image image;
        image.width = width;
        image.height = height;
        image.rgba.assign(rgba, rgba + (size_t)width * height);
        return SavePng(path, image) ? (int)CPT_OK : Fail(CPT_ERR_IO, std::string("cannot write '") + path + "'");
    });
}

int cpt_scene_object_map(const cpt_scene *s, uint32_t index, int which, uint32_t *width, uint32_t *height, uint32_t *rgba, size_t capacity) {
    SceneObject *obj = s ? MeshObject(s, index) : nullptr;
    if (!obj || !width || !height || which < 0 || which > 2) return Fail(CPT_ERR_INVALID, "cpt_scene_object_map: bad argument");
    const Material *mat = obj->GetComponent<MaterialComponent>()->GetMaterial();
    const std::shared_ptr<Image> &image = which == 0 ? mat->diffuse_image : (which == 1 ? mat->specular_image : mat->emission_image);
    *width = image ? image->width : 0;
    *height = image ? image->height : 0;
    if (image && rgba) {
        if (capacity < image->rgba.size()) return Fail(CPT_ERR_INVALID, "cpt_scene_object_map: buffer too small");
        memcpy(rgba, image->rgba.data(), image->rgba.size() * sizeof(uint32_t));
    }
    return CPT_OK;
}

int cpt_write_exr(const char *path, const float *rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba) return Fail(CPT_ERR_INVALID, "cpt_write_exr: null argument");
    return WriteExrRgba(path, rgba, width, height) ? CPT_OK : Fail(CPT_ERR_IO, std::string("cannot write '") + path + "'");
}

}  // extern "C"
