#include <array>
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
// The reference composes object transforms with glm 0.9.9.8 in fp32 and the traversal multiplies
// rays by the result, so the operation order of each glm routine is kept: a matrix that differs
// in the last bit moves hits.
namespace {
using Cols = std::array<V4, 4>;

V4 axpy3(const Cols &m, float a, float b, float c) {     // m[0]*a + m[1]*b + m[2]*c, left to right
    return m[0] * a + m[1] * b + m[2] * c;
}

// glm::translate, thirdparty/glm/glm/ext/matrix_transform.inl:9-15
void Translate(Cols &m, V3 v) { m[3] = axpy3(m, v.x, v.y, v.z) + m[3]; }

// glm::rotate about a coordinate axis, matrix_transform.inl:17-47 (normalize() of a unit axis is exact)
void Rotate(Cols &m, float degrees, int axis_index) {
    const float angle = degrees * static_cast<float>(0.01745329251994329576923690768489);    // glm::radians
    const float c = std::cos(angle), s = std::sin(angle);
    float axis[3] = {0.0f, 0.0f, 0.0f};
    axis[axis_index] = 1.0f;
    const float inv_len = cpt::inversesqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    for (float &a : axis) a = a * inv_len;
    float temp[3];
    for (int i = 0; i < 3; i++) temp[i] = (1.0f - c) * axis[i];
    float rot[3][3];
    rot[0][0] = c + temp[0] * axis[0];
    rot[0][1] = temp[0] * axis[1] + s * axis[2];
    rot[0][2] = temp[0] * axis[2] - s * axis[1];
    rot[1][0] = temp[1] * axis[0] - s * axis[2];
    rot[1][1] = c + temp[1] * axis[1];
    rot[1][2] = temp[1] * axis[2] + s * axis[0];
    rot[2][0] = temp[2] * axis[0] + s * axis[1];
    rot[2][1] = temp[2] * axis[1] - s * axis[0];
    rot[2][2] = c + temp[2] * axis[2];
    const Cols in = m;
    for (int i = 0; i < 3; i++) m[i] = axpy3(in, rot[i][0], rot[i][1], rot[i][2]);
}

// glm::scale, matrix_transform.inl:77-85
void Scale(Cols &m, V3 v) { m[0] = m[0] * v.x; m[1] = m[1] * v.y; m[2] = m[2] * v.z; }
}  // namespace

M4 SceneObject::GetTransform() const {            // src/scene/core.cpp:109-117
    const M4 id = cpt::identity();
    Cols m = {id.c[0], id.c[1], id.c[2], id.c[3]};
    Translate(m, translate);
    Rotate(m, rotate.z, 2);
    Rotate(m, rotate.y, 1);
    Rotate(m, rotate.x, 0);
    Scale(m, scale);
    M4 out;
    for (int i = 0; i < 4; i++) out.c[i] = m[i];
    return out;
}

// glm::inverse(mat4), thirdparty/glm/glm/detail/func_matrix.inl:294-352, which is what
// src/pathtracer/pathtracer.cpp:113 stores next to the transform: eighteen 2x2 minors of the lower
// rows, cofactor columns with alternating signs, determinant from the first column, everything in
// fp32 in glm's order.
M4 InverseTransform(const M4 &mat) {
    float m[4][4];      // m[column][row]
    for (int c = 0; c < 4; c++) { m[c][0] = mat.c[c].x; m[c][1] = mat.c[c].y; m[c][2] = mat.c[c].z; m[c][3] = mat.c[c].w; }
    auto minor2 = [&m](int ca, int ra, int cb, int rb) { return m[ca][ra] * m[cb][rb] - m[cb][ra] * m[ca][rb]; };
    // coef[g] = {minor of columns (2,3), (1,3), (1,2)} for the row pair of group g
    static const int kRows[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
    float fac[6][4];
    for (int g = 0; g < 6; g++) {
        const int ra = kRows[g][0], rb = kRows[g][1];
        const float c23 = minor2(2, ra, 3, rb), c13 = minor2(1, ra, 3, rb), c12 = minor2(1, ra, 2, rb);
        fac[g][0] = c23; fac[g][1] = c23; fac[g][2] = c13; fac[g][3] = c12;
    }
    float vec[4][4];
    for (int r = 0; r < 4; r++) { vec[r][0] = m[1][r]; vec[r][1] = m[0][r]; vec[r][2] = m[0][r]; vec[r][3] = m[0][r]; }
    // column k of the adjugate: the three rows other than k, against the three factor groups that do not involve row k
    static const int kTerms[4][3][2] = {{{1, 0}, {2, 1}, {3, 2}}, {{0, 0}, {2, 3}, {3, 4}}, {{0, 1}, {1, 3}, {3, 5}}, {{0, 2}, {1, 4}, {2, 5}}};
    float inv[4][4];
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) {
            const float a = vec[kTerms[k][0][0]][j] * fac[kTerms[k][0][1]][j];
            const float b = vec[kTerms[k][1][0]][j] * fac[kTerms[k][1][1]][j];
            const float c = vec[kTerms[k][2][0]][j] * fac[kTerms[k][2][1]][j];
            const float sign = ((k + j) & 1) ? -1.0f : 1.0f;
            inv[k][j] = (a - b + c) * sign;
        }
    const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    M4 out;
    for (int c = 0; c < 4; c++) out.c[c] = V4{inv[c][0] * one_over_det, inv[c][1] * one_over_det, inv[c][2] * one_over_det, inv[c][3] * one_over_det};
    return out;
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
