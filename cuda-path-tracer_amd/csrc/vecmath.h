// Small fixed-order vector math shared by host and gfx950 device code.
//
// Every operation spells out its evaluation order because first-hit IDs must be bit-exact
// against the reference, whose arithmetic is glm 0.9.9.8 under IEEE fp32 without contraction:
//   dot(vec3)  = x + y + z, left to right      (thirdparty/glm/glm/detail/func_geometric.inl:48-55)
//   dot(vec4)  = (x + y) + (z + w)             (func_geometric.inl:58-66)
//   cross                                       (func_geometric.inl:68-79)
//   normalize  = v * (1 / sqrt(dot(v, v)))      (func_geometric.inl:82-90)
//   mat4 * vec4 = (m0*x + m1*y) + (m2*z + m3*w) (thirdparty/glm/glm/detail/type_mat4x4.inl:561-572)
// All translation units are compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CPT_HD __host__ __device__ inline
#define CPT_D __device__ inline
#else
#define CPT_HD inline
#define CPT_D inline
#endif

namespace cpt {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { V4 c[4]; };   // column major: c[i] is column i

CPT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
CPT_HD V3 v3(float s) { return V3{s, s, s}; }
CPT_HD V4 v4(V3 v, float w) { return V4{v.x, v.y, v.z, w}; }
CPT_HD V3 xyz(V4 v) { return V3{v.x, v.y, v.z}; }

CPT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
CPT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
CPT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
CPT_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
CPT_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
CPT_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
CPT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
CPT_HD bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
CPT_HD bool operator!=(V3 a, V3 b) { return !(a == b); }
CPT_HD bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
CPT_HD V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
CPT_HD V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
CPT_HD V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
CPT_HD V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
CPT_HD V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }

CPT_HD float dot(V3 a, V3 b) { V3 t = a * b; return t.x + t.y + t.z; }
CPT_HD float dot4(V4 a, V4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
CPT_HD V3 cross(V3 x, V3 y) { return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
CPT_HD float inversesqrt(float x) { return 1.0f / sqrtf(x); }
CPT_HD V3 normalize(V3 v) { return v * inversesqrt(dot(v, v)); }
CPT_HD V4 normalize4(V4 v) { return v * inversesqrt(dot4(v, v)); }
CPT_HD float length(V3 v) { return sqrtf(dot(v, v)); }
// glm::min / glm::max pick the second argument only on a strict comparison
CPT_HD float gmin(float a, float b) { return (b < a) ? b : a; }
CPT_HD float gmax(float a, float b) { return (a < b) ? b : a; }
CPT_HD V3 gmin3(V3 a, V3 b) { return V3{gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
CPT_HD V3 gmax3(V3 a, V3 b) { return V3{gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

CPT_HD V4 mul(const M4 &m, V4 v) {
    V4 mul0 = m.c[0] * v.x, mul1 = m.c[1] * v.y;
    V4 add0 = mul0 + mul1;
    V4 mul2 = m.c[2] * v.z, mul3 = m.c[3] * v.w;
    V4 add1 = mul2 + mul3;
    return add0 + add1;
}
// glm::mat3(m) * v
CPT_HD V3 mul3(const M4 &m, V3 v) {
    return V3{m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
              m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
              m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
CPT_HD M4 transpose(const M4 &m) {
    M4 r;
    r.c[0] = V4{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x};
    r.c[1] = V4{m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y};
    r.c[2] = V4{m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z};
    r.c[3] = V4{m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w};
    return r;
}
inline M4 identity() {
    M4 r;
    r.c[0] = V4{1, 0, 0, 0}; r.c[1] = V4{0, 1, 0, 0}; r.c[2] = V4{0, 0, 1, 0}; r.c[3] = V4{0, 0, 0, 1};
    return r;
}

constexpr float kPi = 3.14159265359f;
constexpr float k2Pi = 2.0f * kPi;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kInv2Pi = 1.0f / k2Pi;

// Rec.601 luma with the first two products and both sums in double, as
// src/kernels/basic/prelude.cuh:24-26 writes it (0.299 and 0.587 are double literals);
// the value feeds Russian roulette and lobe selection, so the width matters.
CPT_HD float luminance(V3 c) {
    return (float)((double)c.x * 0.299 + (double)c.y * 0.587 + (double)(c.z * 0.114f));
}

}  // namespace cpt
