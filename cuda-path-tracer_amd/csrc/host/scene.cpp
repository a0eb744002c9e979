#include "scene.hpp"

#include <cmath>

#include "pathtracer.hpp"

using cpt::M4;
using cpt::V2;
using cpt::V3;
using cpt::V4;

// ---------------------------------------------------------------- Bbox (src/scene/bbox.cpp)
V3 Bbox::Centroid() const { return (pmin + pmax) * 0.5f; }
float Bbox::Extent() const { return cpt::length(pmax - pmin); }
void Bbox::Empty() {
    pmin = V3{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    pmax = V3{-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
}
Bbox Bbox::TransformBy(const M4 &mat) const {     // bbox.cpp:18-37: eight corners, pairwise min/max tree
    V3 p[8];
    const float xs[2] = {pmin.x, pmax.x}, ys[2] = {pmin.y, pmax.y}, zs[2] = {pmin.z, pmax.z};
    for (int i = 0; i < 8; i++) p[i] = cpt::xyz(cpt::mul(mat, V4{xs[i & 1], ys[(i >> 1) & 1], zs[(i >> 2) & 1], 1.0f}));
    Bbox r;
    r.pmin = cpt::gmin3(cpt::gmin3(cpt::gmin3(p[0], p[1]), cpt::gmin3(p[2], p[3])), cpt::gmin3(cpt::gmin3(p[4], p[5]), cpt::gmin3(p[6], p[7])));
    r.pmax = cpt::gmax3(cpt::gmax3(cpt::gmax3(p[0], p[1]), cpt::gmax3(p[2], p[3])), cpt::gmax3(cpt::gmax3(p[4], p[5]), cpt::gmax3(p[6], p[7])));
    return r;
}
void Bbox::Merge(const Bbox &rhs) { pmin = cpt::gmin3(pmin, rhs.pmin); pmax = cpt::gmax3(pmax, rhs.pmax); }
void Bbox::Merge(const V3 &p) { pmin = cpt::gmin3(pmin, p); pmax = cpt::gmax3(pmax, p); }
bool Bbox::IntersectWith(const Bbox &rhs) const {
    return pmin.x <= rhs.pmax.x && pmax.x >= rhs.pmin.x && pmin.y <= rhs.pmax.y && pmax.y >= rhs.pmin.y && pmin.z <= rhs.pmax.z &&
           pmax.z >= rhs.pmin.z;
}

// ---------------------------------------------------------------- Mesh (src/scene/mesh.cpp:7-82)
void Mesh::SetPositions(std::vector<V3> &&positions) {
    positions_ = std::move(positions);
    bbox_.Empty();
    for (const V3 &p : positions_) bbox_.Merge(p);
}

void Mesh::CalcNormals() {
    normals_.assign(positions_.size(), V3{0.0f, 0.0f, 0.0f});
    for (size_t i = 0; i + 2 < indices_.size(); i += 3) {
        V3 p0 = positions_[indices_[i]], p1 = positions_[indices_[i + 1]], p2 = positions_[indices_[i + 2]];
        V3 n = cpt::normalize(cpt::cross(p1 - p0, p2 - p0));
        normals_[indices_[i]] = normals_[indices_[i]] + n;
        normals_[indices_[i + 1]] = normals_[indices_[i + 1]] + n;
        normals_[indices_[i + 2]] = normals_[indices_[i + 2]] + n;
    }
    for (V3 &n : normals_) n = cpt::normalize(n);
}

void Mesh::CalcTexcoords() {
    texcoords_.resize(positions_.size());
    const double pi = 3.141592653589793238462643383279502884, inv_pi = 0.318309886183790671537767526745028724;
    for (size_t i = 0; i < positions_.size(); i++) {
        V3 vec = cpt::normalize(positions_[i] - bbox_.Centroid());
        float theta = std::acos(vec.y);
        float phi = std::atan2(vec.z, vec.x);
        if (phi < 0.0f) phi += 2.0f * pi;
        texcoords_[i] = V2{(float)(phi * inv_pi * 0.5f), (float)(1.0f - theta * inv_pi)};
    }
}

// ---------------------------------------------------------------- Material (src/scene/material.cpp:57-59)
bool Material::IsEmissive() const { return cpt::dot(emission, emission) > 0.0f; }

// ---------------------------------------------------------------- SceneObject / Scene
namespace {
M4 matmul(const M4 &a, const M4 &b) {
    M4 r;
    for (int c = 0; c < 4; c++) r.c[c] = cpt::mul(a, b.c[c]);
    return r;
}
M4 rotation(float degrees, V3 axis) {
    float a = degrees * 0.01745329251994329576923690768489f;
    float c = std::cos(a), s = std::sin(a);
    V3 t = axis * (1.0f - c);
    M4 r = cpt::identity();
    r.c[0] = V4{c + t.x * axis.x, t.x * axis.y + s * axis.z, t.x * axis.z - s * axis.y, 0.0f};
    r.c[1] = V4{t.y * axis.x - s * axis.z, c + t.y * axis.y, t.y * axis.z + s * axis.x, 0.0f};
    r.c[2] = V4{t.z * axis.x + s * axis.y, t.z * axis.y - s * axis.x, c + t.z * axis.z, 0.0f};
    return r;
}
}  // namespace

M4 SceneObject::GetTransform() const {            // src/scene/core.cpp:109-117
    M4 trans = cpt::identity();
    M4 t = cpt::identity();
    t.c[3] = V4{translate.x, translate.y, translate.z, 1.0f};
    trans = matmul(trans, t);
    trans = matmul(trans, rotation(rotate.z, V3{0.0f, 0.0f, 1.0f}));
    trans = matmul(trans, rotation(rotate.y, V3{0.0f, 1.0f, 0.0f}));
    trans = matmul(trans, rotation(rotate.x, V3{1.0f, 0.0f, 0.0f}));
    M4 s = cpt::identity();
    s.c[0].x = scale.x; s.c[1].y = scale.y; s.c[2].z = scale.z;
    return matmul(trans, s);
}

std::shared_ptr<SceneObject> Scene::AddObject(std::string_view name) {
    objects_.push_back(std::make_shared<SceneObject>(name));
    return objects_.back();
}

void Scene::Update() {                            // src/scene/core.cpp:87-91
    for (auto &obj : objects_)
        if (auto *c = obj->GetComponent<CameraComponent>()) c->Update();
    for (auto &obj : objects_)
        if (auto *p = obj->GetComponent<PathTracer>()) p->Update();
}
