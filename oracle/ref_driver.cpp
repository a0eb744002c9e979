// TEST INFRASTRUCTURE ONLY.  Thin C-ABI driver around the REFERENCE's own header-only
// device code, compiled for the host where the sources lie (/root/reference); nothing
// from the reference is copied into this repository and the built library lands in
// oracle/_ref/ (git-ignored).
//
// Buildable with plain g++ (every CU_* qualifier is empty without __CUDACC__,
// src/kernels/basic/prelude.cuh:3-13):
//   src/kernels/accel/accel.cuh          BboxIntersect, AccelBottom/AccelTop Intersect/Occlude
//   src/kernels/geometry/*.cuh           TriMesh::Intersect/GetVertex/SampleVertex, Geometry
//   src/kernels/bsdf/*.cuh               all five BSDFs
//   src/kernels/sampler/*.cuh            TEA seed + PCG step
//   src/kernels/camera/*.cuh             PinholeCamera::SampleRay
//   src/kernels/basic/frame.cuh          Frame
// NOT buildable here (need the CUDA toolkit: <texture_types.h>, <<<>>> launches, thrust,
// cuda_runtime): material/*.cuh and everything that includes it (scene/, light/,
// integrator/), accel_build.cu, path.cu.
#include <cmath>
#include <math.h>
#include <stdlib.h>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <limits>
#include <vector>

#include <glm/gtc/matrix_transform.hpp>   // glm::translate / rotate / scale, as src/scene/core.cpp includes them
#include "kernels/accel/accel.cuh"
#include "kernels/bsdf/bsdf.cuh"
#include "kernels/sampler/sampler.cuh"
#include "kernels/camera/camera.cuh"

namespace {

struct RefMesh {
    std::vector<glm::vec3> positions, normals;
    std::vector<glm::vec2> texcoords;
    std::vector<uint32_t> indices;
    std::vector<kernel::AccelNode> nodes;
    std::vector<kernel::Bbox> bboxes;
    kernel::TriMesh trimesh;
    kernel::AccelBottom accel;
};

struct RefScene {
    std::vector<RefMesh *> meshes;
    std::vector<kernel::AccelTop::Instance> instances;
    std::vector<kernel::AccelNode> top_nodes;
    std::vector<kernel::Bbox> top_bboxes;
    kernel::AccelTop top;
    ~RefScene() { for (auto m : meshes) delete m; }
};

kernel::Ray MakeRay(const float *p) {
    kernel::Ray r(glm::vec3(p[0], p[1], p[2]), glm::vec3(p[4], p[5], p[6]));
    r.tmin = p[3];
    r.tmax = p[7];
    return r;
}

}  // namespace

extern "C" {

void ref_sizes(uint32_t *out8) {
    out8[0] = sizeof(kernel::Bbox); out8[1] = sizeof(kernel::AccelNode); out8[2] = sizeof(kernel::Ray);
    out8[3] = sizeof(kernel::Bsdf); out8[4] = sizeof(kernel::AccelTop::Instance); out8[5] = sizeof(kernel::Vertex);
    out8[6] = sizeof(kernel::TriMesh); out8[7] = sizeof(kernel::AccelHitInfo);
}

void *ref_scene_create() { return new RefScene(); }
void ref_scene_destroy(void *h) { delete static_cast<RefScene *>(h); }

// nodes/bboxes: a finished BLAS (2*ntris-1 entries) in the reference's own layout
int ref_scene_add_mesh(void *h, const float *pos, const float *nrm, const float *uv, uint32_t nverts, const uint32_t *idx, uint32_t ntris,
                       const uint32_t *nodes, const float *bboxes) {
    RefScene &sc = *static_cast<RefScene *>(h);
    RefMesh *m = new RefMesh();
    m->positions.resize(nverts); m->normals.resize(nverts); m->texcoords.resize(nverts);
    memcpy(m->positions.data(), pos, 12 * (size_t)nverts);
    memcpy(m->normals.data(), nrm, 12 * (size_t)nverts);
    memcpy(m->texcoords.data(), uv, 8 * (size_t)nverts);
    m->indices.assign(idx, idx + 3 * (size_t)ntris);
    size_t nn = 2 * (size_t)ntris - 1;
    m->nodes.resize(nn); m->bboxes.resize(nn);
    memcpy(m->nodes.data(), nodes, sizeof(kernel::AccelNode) * nn);
    memcpy(m->bboxes.data(), bboxes, sizeof(kernel::Bbox) * nn);
    m->trimesh = kernel::TriMesh{m->positions.data(), m->normals.data(), m->texcoords.data(), m->indices.data(), ntris};
    m->accel.nodes = m->nodes.data();
    m->accel.bboxes = m->bboxes.data();
    m->accel.geometry.type = kernel::Geometry::Type::eTriMesh;
    m->accel.geometry.ptr = &m->trimesh;
    sc.meshes.push_back(m);
    return (int)sc.meshes.size() - 1;
}

// SceneObject::GetTransform (src/scene/core.cpp:109-117) and the inverse PathTracer::BuildAccel
// stores with it (src/pathtracer/pathtracer.cpp:112-114), through the reference's own glm
void ref_trs_matrix(const float *t3, const float *r3, const float *s3, float *out16, float *out_inv16) {
    glm::vec3 translate(t3[0], t3[1], t3[2]), rotate(r3[0], r3[1], r3[2]), scale(s3[0], s3[1], s3[2]);
    glm::mat4 trans(1.0f);
    trans = glm::translate(trans, translate);
    trans = glm::rotate(trans, glm::radians(rotate.z), glm::vec3(0.0f, 0.0f, 1.0f));
    trans = glm::rotate(trans, glm::radians(rotate.y), glm::vec3(0.0f, 1.0f, 0.0f));
    trans = glm::rotate(trans, glm::radians(rotate.x), glm::vec3(1.0f, 0.0f, 0.0f));
    trans = glm::scale(trans, scale);
    glm::mat4 inv = glm::inverse(trans);
    memcpy(out16, &trans, 64);
    memcpy(out_inv16, &inv, 64);
}

int ref_scene_add_instance(void *h, int mesh, const float *transform16, const float *transform_inv16) {
    RefScene &sc = *static_cast<RefScene *>(h);
    kernel::AccelTop::Instance inst;
    inst.accel = &sc.meshes[mesh]->accel;
    memcpy(&inst.transform, transform16, 64);
    memcpy(&inst.transform_inv, transform_inv16, 64);
    sc.instances.push_back(inst);
    return (int)sc.instances.size() - 1;
}

void ref_scene_set_top(void *h, const uint32_t *nodes, const float *bboxes) {
    RefScene &sc = *static_cast<RefScene *>(h);
    size_t nn = 2 * sc.instances.size() - 1;
    sc.top_nodes.resize(nn); sc.top_bboxes.resize(nn);
    memcpy(sc.top_nodes.data(), nodes, sizeof(kernel::AccelNode) * nn);
    memcpy(sc.top_bboxes.data(), bboxes, sizeof(kernel::Bbox) * nn);
    sc.top.nodes = sc.top_nodes.data();
    sc.top.bboxes = sc.top_bboxes.data();
    sc.top.instances = sc.instances.data();
}

// AccelTop::Intersect (src/kernels/accel/accel.cuh:124-150) on n rays
void ref_first_hit(void *h, const float *rays, uint64_t n, uint32_t *inst_out, uint32_t *prim_out, float *t_out, float *uv_out) {
    RefScene &sc = *static_cast<RefScene *>(h);
    for (uint64_t i = 0; i < n; i++) {
        kernel::Ray r = MakeRay(rays + 8 * i);
        kernel::AccelHitInfo hit;
        bool ok = sc.top.Intersect(r, hit);
        inst_out[i] = ok ? hit.instance_id : 0xffffffffu;
        prim_out[i] = ok ? hit.primitive_id : 0xffffffffu;
        if (t_out) t_out[i] = ok ? r.tmax : rays[8 * i + 7];
        if (uv_out) { uv_out[2 * i] = ok ? hit.attribs.x : 0.0f; uv_out[2 * i + 1] = ok ? hit.attribs.y : 0.0f; }
    }
}

// AccelTop::Occlude (accel.cuh:152-173)
void ref_occlude(void *h, const float *rays, uint64_t n, uint8_t *out) {
    RefScene &sc = *static_cast<RefScene *>(h);
    for (uint64_t i = 0; i < n; i++) {
        kernel::Ray r = MakeRay(rays + 8 * i);
        out[i] = sc.top.Occlude(r) ? 1 : 0;
    }
}

int ref_bbox_intersect(const float *pmin3, const float *pmax3, const float *ray8) {
    kernel::Ray r = MakeRay(ray8);
    return kernel::BboxIntersect(glm::vec3(pmin3[0], pmin3[1], pmin3[2]), glm::vec3(pmax3[0], pmax3[1], pmax3[2]), r) ? 1 : 0;
}

// TriMesh::GetVertex / SampleVertex (src/kernels/geometry/trimesh.cuh:45-80), untransformed
void ref_mesh_vertex(void *h, int mesh, int sample, uint32_t prim, float a, float b, float *out9) {
    RefScene &sc = *static_cast<RefScene *>(h);
    kernel::Vertex v = sample ? sc.meshes[mesh]->trimesh.SampleVertex(glm::vec2(a, b))
                              : sc.meshes[mesh]->trimesh.GetVertex(prim, glm::vec2(a, b));
    out9[0] = v.position.x; out9[1] = v.position.y; out9[2] = v.position.z;
    out9[3] = v.normal.x; out9[4] = v.normal.y; out9[5] = v.normal.z;
    out9[6] = v.texcoord.x; out9[7] = v.texcoord.y; out9[8] = v.pdf;
}

uint32_t ref_tea(uint32_t x, uint32_t y) { return kernel::SamplerState::Create(x, y).state; }
void ref_next1d_seq(uint32_t state, uint32_t n, float *out) {
    kernel::SamplerState s{kernel::SamplerState::Type::eRandom, state};
    for (uint32_t i = 0; i < n; i++) out[i] = s.Next1D();
}

static kernel::Bsdf MakeBsdf(int type, const float *p14) {
    // p14: diffuse[3] specular[3] transmittance[3] shininess s_norm ior roughness opacity
    kernel::Bsdf b{};
    b.type = static_cast<kernel::Bsdf::Type>(type);
    glm::vec3 diffuse(p14[0], p14[1], p14[2]), specular(p14[3], p14[4], p14[5]), transmittance(p14[6], p14[7], p14[8]);
    switch (b.type) {
        case kernel::Bsdf::Type::eLambert: { auto d = reinterpret_cast<kernel::LambertBsdf *>(b.data); d->color = diffuse; break; }
        case kernel::Bsdf::Type::ePhong: {
            auto d = reinterpret_cast<kernel::PhongBsdf *>(b.data);
            d->diffuse = diffuse; d->specular = specular; d->shininess = p14[9]; d->s_norm = p14[10]; break;
        }
        case kernel::Bsdf::Type::eBlinnPhong: {
            auto d = reinterpret_cast<kernel::BlinnPhongBsdf *>(b.data);
            d->diffuse = diffuse; d->specular = specular; d->shininess = p14[9]; d->s_norm = p14[10]; break;
        }
        case kernel::Bsdf::Type::eMicrofacet: {
            auto d = reinterpret_cast<kernel::MicrofacetBsdf *>(b.data);
            d->diffuse = diffuse; d->specular = specular; d->transmittance = transmittance;
            d->ior = p14[11]; d->roughness = p14[12]; d->opacity = p14[13]; break;
        }
        case kernel::Bsdf::Type::eGlass: {
            // same slot convention as the restatement: reflectance = p14[0..2], transmittance = p14[3..5]
            auto d = reinterpret_cast<kernel::GlassBsdf *>(b.data);
            d->reflectance = diffuse; d->transmittance = specular; d->ior = p14[11]; break;
        }
    }
    return b;
}

// Bsdf::Sample (src/kernels/bsdf/bsdf.cuh:39-52); out9: wi[3] pdf weight[3] lobe_type lobe_dir
void ref_bsdf_sample(int type, const float *p14, const float *wo3, float r1, float r2x, float r2y, float *out9) {
    kernel::Bsdf b = MakeBsdf(type, p14);
    kernel::BsdfSample s = b.Sample(glm::vec3(wo3[0], wo3[1], wo3[2]), r1, glm::vec2(r2x, r2y));
    out9[0] = s.wi.x; out9[1] = s.wi.y; out9[2] = s.wi.z; out9[3] = s.pdf;
    out9[4] = s.weight.x; out9[5] = s.weight.y; out9[6] = s.weight.z;
    out9[7] = (float)(int)s.lobe.type; out9[8] = (float)(int)s.lobe.dir;
}

// Bsdf::Pdf / Bsdf::Eval (bsdf.cuh:54-82); out4: pdf eval[3]
void ref_bsdf_eval(int type, const float *p14, const float *wo3, const float *wi3, float *out4) {
    kernel::Bsdf b = MakeBsdf(type, p14);
    glm::vec3 wo(wo3[0], wo3[1], wo3[2]), wi(wi3[0], wi3[1], wi3[2]);
    out4[0] = b.Pdf(wo, wi);
    glm::vec3 e = b.Eval(wo, wi);
    out4[1] = e.x; out4[2] = e.y; out4[3] = e.z;
}

// PinholeCamera::SampleRay (src/kernels/camera/pinhole.cuh:16-22); cam18 = pos fx fy fz x_dir y_dir
void ref_camera_ray(const float *cam18, float aspect, float u, float v, float *ray8) {
    kernel::PinholeCamera cam;
    memcpy(&cam, cam18, sizeof(cam));
    kernel::Ray r = cam.SampleRay(aspect, glm::vec2(u, v), glm::vec2(0.0f));
    ray8[0] = r.origin.x; ray8[1] = r.origin.y; ray8[2] = r.origin.z; ray8[3] = r.tmin;
    ray8[4] = r.direction.x; ray8[5] = r.direction.y; ray8[6] = r.direction.z; ray8[7] = r.tmax;
}

// Frame(normal) (src/kernels/basic/frame.cuh:13-20)
void ref_frame(const float *n3, float *out9) {
    kernel::Frame f(glm::vec3(n3[0], n3[1], n3[2]));
    out9[0] = f.x.x; out9[1] = f.x.y; out9[2] = f.x.z; out9[3] = f.y.x; out9[4] = f.y.y; out9[5] = f.y.z; out9[6] = f.z.x; out9[7] = f.z.y; out9[8] = f.z.z;
}

float ref_luminance(float r, float g, float b) { return kernel::Luminance(glm::vec3(r, g, b)); }

}  // extern "C"
