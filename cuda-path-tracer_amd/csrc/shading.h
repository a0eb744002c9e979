// Bounce-loop building blocks for gfx950 (device code): counter-based per-pixel RNG, camera,
// shading surface, area-light sampling and the five BSDFs.  The expression order follows the
// reference so that, given the same random numbers, paths agree up to the last-ulp differences of
// sinf/cosf/powf between math libraries.  Wherever the reference leaves the order of two random
// draws to the compiler (argument lists), the draws are taken left to right.
#pragma once
#include "device_types.h"

namespace cpt {

// ------------------------------------------------------------------ RNG
// TEA-16 seed from (pixel index, sample index), src/kernels/sampler/random.cuh:9-19
CPT_HD uint32_t rand_init_tea(uint32_t x, uint32_t y) {
    uint32_t v0 = x, v1 = y, s0 = 0;
    for (uint32_t n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
// PCG-RXS-M-XS step, random.cuh:21-26
CPT_HD uint32_t rand_pcg(uint32_t &state) {
    uint32_t prev = state;
    state = state * 747796405u + 2891336453u;
    uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
    return (word >> 22u) ^ word;
}
struct Sampler {
    uint32_t state;
    CPT_HD float next1d() { return (float)rand_pcg(state) / 4294967296.0f; }   // random.cuh:38-41 (may return 1.0f)
    CPT_HD V2 next2d() { float a = next1d(); float b = next1d(); return V2{a, b}; }
};

// ------------------------------------------------------------------ frame (src/kernels/basic/frame.cuh)
struct Frame {
    V3 x, y, z;
    CPT_HD static Frame from_normal(V3 n) {                 // :13-20
        Frame f;
        f.z = n;
        float sign = n.z > 0.0f ? 1.0f : -1.0f;
        float a = -1.0f / (sign + n.z);
        float b = n.x * n.y * a;
        f.x = V3{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
        f.y = V3{b, sign + n.y * n.y * a, -n.y};
        return f;
    }
    CPT_HD V3 to_local(V3 p) const { return V3{dot(x, p), dot(y, p), dot(z, p)}; }            // :25-28 (origin 0, w 0)
    CPT_HD V3 to_world(V3 p) const { return x * p.x + y * p.y + z * p.z + v3(0.0f); }         // :30-32
};

struct Vertex { V3 position, normal; float pdf; V2 texcoord; };

// src/kernels/geometry/common.cuh:15-20
CPT_HD void vertex_transform_pdf(Vertex &v, const M4 &transform) {
    Frame f = Frame::from_normal(v.normal);
    V3 t = mul3(transform, f.x);
    V3 b = mul3(transform, f.y);
    v.pdf /= length(cross(t, b));
}

// Mesh arrays are global memory; saying so in the pointer type turns the fetches below into global
// loads (through the generic pointers of MeshDev they would be FLAT accesses).
typedef const float __attribute__((address_space(1))) *GlobalFloats;
typedef const uint32_t __attribute__((address_space(1))) *GlobalWords;
CPT_D V3 load_v3(const V3 *base, uint32_t i) { GlobalFloats p = (GlobalFloats)(base + i); return V3{p[0], p[1], p[2]}; }
CPT_D V2 load_v2(const V2 *base, uint32_t i) { GlobalFloats p = (GlobalFloats)(base + i); return V2{p[0], p[1]}; }

// src/kernels/geometry/trimesh.cuh:45-72; texture coordinates are fetched only for materials that read them
CPT_D Vertex get_vertex(const MeshDev &m, uint32_t prim, float ax, float ay, bool want_texcoord = false) {
    GlobalWords idx = (GlobalWords)(m.indices + prim * 3);
    uint32_t i0 = idx[0], i1 = idx[1], i2 = idx[2];
    Vertex vert;
    V3 pos0 = load_v3(m.positions, i0), pos1 = load_v3(m.positions, i1), pos2 = load_v3(m.positions, i2);
    V3 e1 = pos1 - pos0, e2 = pos2 - pos0;
    float area = length(cross(e1, e2)) * 0.5f;
    vert.pdf = 1.0f / area / (float)m.num_triangles;
    vert.position = pos0 + ax * (pos1 - pos0) + ay * (pos2 - pos0);
    V3 n0 = load_v3(m.normals, i0), n1 = load_v3(m.normals, i1), n2 = load_v3(m.normals, i2);
    vert.normal = normalize(n0 + ax * (n1 - n0) + ay * (n2 - n0));
    vert.texcoord = V2{0.0f, 0.0f};
    if (want_texcoord) {
        V2 t0 = load_v2(m.texcoords, i0), t1 = load_v2(m.texcoords, i1), t2 = load_v2(m.texcoords, i2);
        vert.texcoord = t0 + ax * (t1 - t0) + ay * (t2 - t0);
    }
    return vert;
}

// float -> uint32 with the saturating behaviour of the hardware conversion
CPT_HD uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// trimesh.cuh:74-80: uniform triangle, then uniform point on it
CPT_D Vertex sample_vertex(const MeshDev &m, V2 rand, bool want_texcoord = false) {
    float nt = (float)m.num_triangles;
    uint32_t prim = min(f2u_sat(rand.x * nt), m.num_triangles - 1);
    float u_sqrt = sqrtf(rand.x * nt - (float)prim);
    float u = 1.0f - u_sqrt;
    float v = (1.0f - rand.y) * u_sqrt;
    return get_vertex(m, prim, u, v, want_texcoord);
}

// ------------------------------------------------------------------ BSDFs (src/kernels/bsdf/*.cuh)
struct BsdfSample { V3 wi; float pdf; V3 weight; int lobe; };

CPT_HD V3 cosine_hemisphere_sample(V2 rand) {               // bsdf/common.cuh:28-36
    float phi = rand.x * k2Pi;
    float sin_phi = sinf(phi), cos_phi = cosf(phi);
    float sin_theta = sqrtf(rand.y), cos_theta = sqrtf(1.0f - rand.y);
    return V3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
}
CPT_HD V3 reflect(V3 i, V3 n) { return 2.0f * dot(i, n) * n - i; }          // :38-40
CPT_HD V3 refract(V3 i, V3 n, float ior) {                                  // :41-51
    float cos_i = dot(i, n);
    float eta = cos_i >= 0.0f ? 1.0f / ior : ior;
    float cos_t_sqr = 1.0f - (1.0f - cos_i * cos_i) * eta * eta;
    if (cos_t_sqr < 0.0f) return v3(0.0f);
    float cos_t = sqrtf(cos_t_sqr);
    float n_scale = eta * fabsf(cos_i) - cos_t;
    return (cos_i >= 0.0f ? n_scale : -n_scale) * n - eta * i;
}
CPT_HD V3 reflect_half(V3 i, V3 o) { V3 h = normalize(i + o); return h.z >= 0.0f ? h : -h; }   // :53-56
CPT_HD V3 refract_half(V3 i, V3 o, float ior) {                                                  // :57-61
    float eta = i.z >= 0.0f ? 1.0f / ior : ior;
    V3 h = normalize(eta * i + o);
    return h.z >= 0.0f ? h : -h;
}
CPT_HD float pow2(float x) { return x * x; }
CPT_HD float fresnel(float ior, V3 i, V3 n) {                                                     // :71-84
    float eta = dot(i, n) < 0.0f ? ior : 1.0f / ior;
    V3 r = refract(i, n, ior);
    if (!is_zero(r)) {
        float idotn = fabsf(dot(i, n));
        float tdotn = fabsf(dot(r, n));
        float rs = pow2((idotn - eta * tdotn) / (idotn + eta * tdotn));
        float rp = pow2((tdotn - eta * idotn) / (tdotn + eta * idotn));
        return 0.5f * (rs + rp);
    }
    return 1.0f;
}

CPT_HD BsdfSample empty_sample() { return BsdfSample{v3(0.0f), 0.0f, v3(0.0f), kLobeNone}; }

// ---- Lambert (bsdf/lambert.cuh:12-24)
CPT_HD BsdfSample lambert_sample(const BsdfDev &b, V3 wo, V2 r2) {
    V3 wi = cosine_hemisphere_sample(r2);
    wi.z = copysignf(wi.z, wo.z);
    return BsdfSample{wi, fabsf(wi.z) * kInvPi, b.diffuse, kLobeDiffuse};
}
CPT_HD float lambert_pdf(V3 wo, V3 wi) { return wo.z * wi.z >= 0.0f ? fabsf(wi.z) * kInvPi : 1.0f; }
CPT_HD V3 lambert_eval(const BsdfDev &b, V3 wo, V3 wi) { return wo.z * wi.z >= 0.0f ? b.diffuse * fabsf(wi.z) * kInvPi : v3(0.0f); }

// ---- Blinn-Phong (bsdf/blinn_phong.cuh:15-75)
// pow(h.z, shininess) where the reference computes it.  With a specular colour of zero every use multiplies it by
// a zero, and a finite factor then leaves no trace in the result, bit for bit; it is finite for h.z >= 0.  (For
// h.z < 0 it is a NaN that does reach the result, so it is evaluated.)  The Cornell scenes have Ks = 0 throughout.
CPT_HD float blinn_lobe(const BsdfDev &b, float hz, float specular_pdf) {
    const bool unused = b.specular.x == 0.0f && b.specular.y == 0.0f && b.specular.z == 0.0f && specular_pdf == 0.0f && hz >= 0.0f;
    return unused ? 0.0f : powf(hz, b.shininess);
}
CPT_HD BsdfSample blinn_sample(const BsdfDev &b, V3 wo, float r1, V2 r2) {
    float diffuse_weight = luminance(b.diffuse), specular_weight = luminance(b.specular);
    float diffuse_pdf = diffuse_weight / (diffuse_weight + specular_weight);
    float specular_pdf = 1.0f - diffuse_pdf;
    BsdfSample samp = empty_sample();
    float d, s;
    V3 h;
    if (r1 < diffuse_pdf) {
        samp.wi = cosine_hemisphere_sample(r2);
        samp.wi.z = copysignf(samp.wi.z, wo.z);
        samp.lobe = kLobeDiffuse;
        d = fabsf(samp.wi.z) * kInvPi;
        h = reflect_half(samp.wi, wo);
        s = blinn_lobe(b, h.z, specular_pdf);
    } else {
        float phi = r2.x * k2Pi;
        float sin_phi = sinf(phi), cos_phi = cosf(phi);
        float cos_theta = powf(1.0f - r2.y, 1.0f / (2.0f + b.shininess));
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        h = V3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
        samp.wi = reflect(wo, h);
        samp.lobe = kLobeGlossy;
        s = powf(h.z, b.shininess);
        d = fabsf(samp.wi.z) * kInvPi;
    }
    float half_pdf = s * (2.0f + b.shininess) * kInv2Pi;
    samp.pdf = diffuse_pdf * d + specular_pdf * half_pdf / (4.0f * fabsf(dot(samp.wi, h)));
    samp.weight = (b.diffuse * d + b.specular * s * b.s_norm) / samp.pdf;
    return samp;
}
CPT_HD float blinn_pdf(const BsdfDev &b, V3 wo, V3 wi) {
    if (wo.z * wi.z <= 0.0f) return 1.0f;
    float diffuse_weight = luminance(b.diffuse), specular_weight = luminance(b.specular);
    float diffuse_pdf = diffuse_weight / (diffuse_weight + specular_weight);
    float specular_pdf = 1.0f - diffuse_pdf;
    V3 h = reflect_half(wi, wo);
    float half_pdf = blinn_lobe(b, h.z, specular_pdf) * (2.0f + b.shininess) * kInv2Pi;
    return diffuse_pdf * fabsf(wi.z) * kInvPi + specular_pdf * half_pdf / (4.0f * fabsf(dot(wi, h)));
}
CPT_HD V3 blinn_eval(const BsdfDev &b, V3 wo, V3 wi) {
    if (wo.z * wi.z <= 0.0f) return v3(0.0f);
    V3 h = reflect_half(wi, wo);
    V3 d = b.diffuse * fabsf(wi.z) * kInvPi;
    V3 s = b.specular * blinn_lobe(b, h.z, 0.0f) * b.s_norm;
    return d + s;
}

// ---- Phong (bsdf/phong.cuh:16-71)
CPT_HD float phong_pdf(const BsdfDev &b, V3 wo, V3 wi) {
    if (wo.z * wi.z <= 0.0f) return 1.0f;
    float diffuse_weight = luminance(b.diffuse), specular_weight = luminance(b.specular);
    float diffuse_pdf = diffuse_weight / (diffuse_weight + specular_weight);
    float specular_pdf = 1.0f - diffuse_pdf;
    V3 r = V3{-wo.x, -wo.y, wo.z};
    return diffuse_pdf * fabsf(wi.z) * kInvPi + specular_pdf * powf(dot(r, wi), b.shininess) * b.s_norm;
}
CPT_HD V3 phong_eval(const BsdfDev &b, V3 wo, V3 wi) {
    if (wo.z * wi.z <= 0.0f) return v3(0.0f);
    V3 r = V3{-wo.x, -wo.y, wo.z};
    V3 d = b.diffuse * fabsf(wi.z) * kInvPi;
    V3 s = b.specular * powf(dot(r, wi), b.shininess) * b.s_norm;
    return d + s;
}
CPT_HD BsdfSample phong_sample(const BsdfDev &b, V3 wo, float r1, V2 r2) {
    float diffuse_weight = luminance(b.diffuse), specular_weight = luminance(b.specular);
    float diffuse_pdf = diffuse_weight / (diffuse_weight + specular_weight);
    BsdfSample samp = empty_sample();
    if (r1 < diffuse_pdf) {
        samp.wi = cosine_hemisphere_sample(r2);
        samp.wi.z = copysignf(samp.wi.z, wo.z);
        samp.lobe = kLobeDiffuse;
    } else {
        float phi = r2.x * k2Pi;
        float sin_phi = sinf(phi), cos_phi = cosf(phi);
        float cos_theta = powf(1.0f - r2.y, 1.0f / (1.0f + b.shininess));
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        V3 wi = V3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
        Frame frame = Frame::from_normal(V3{-wo.x, -wo.y, wo.z});
        samp.wi = frame.to_world(wi);
        if (samp.wi.z * wo.x <= 0.0f) { samp.pdf = 0.0f; return samp; }   // phong.cuh:37 tests wo.x, kept as written
        samp.lobe = kLobeGlossy;
    }
    samp.pdf = phong_pdf(b, wo, samp.wi);
    samp.weight = phong_eval(b, wo, samp.wi) / samp.pdf;
    return samp;
}

// ---- Microfacet / GGX (bsdf/microfacet.cuh:9-138)
CPT_HD float ggx_ndf(float ndoth, float a2) { return a2 * kInvPi / pow2(ndoth * ndoth * (a2 - 1.0f) + 1.0f); }
CPT_HD float smith_vis(float ndotv, float ndotl, float a2) {
    float v = ndotl * sqrtf(a2 + (1.0f - a2) * ndotv * ndotv);
    float l = ndotv * sqrtf(a2 + (1.0f - a2) * ndotl * ndotl);
    return 0.5f / (v + l);
}
CPT_HD void micro_weights(const BsdfDev &b, V3 wo, float &sw, float &dw, float &tw) {
    float fr_macro = fresnel(b.ior, wo, v3(0.0f, 0.0f, 1.0f));
    sw = luminance(b.specular) * fr_macro;
    dw = luminance(b.diffuse) * (1.0f - fr_macro) * b.opacity;
    tw = luminance(b.transmittance) * (1.0f - fr_macro) * (1.0f - b.opacity);
    float inv = 1.0f / (sw + dw + tw);
    sw *= inv; dw *= inv; tw *= inv;
}
CPT_HD float micro_pdf(const BsdfDev &b, V3 wo, V3 wi) {
    float sw, dw, tw;
    micro_weights(b, wo, sw, dw, tw);
    float a2 = b.roughness * b.roughness;
    if (wo.z * wi.z <= 0.0f) {
        V3 h = refract_half(wi, wo, b.ior);
        float ndf = ggx_ndf(h.z, a2);
        float eta = wo.z >= 0.0f ? 1.0f / b.ior : b.ior;
        float denom = pow2(eta * dot(wo, h) + dot(wi, h));
        float num = fabsf(dot(wi, h));
        return tw * ndf * num / denom;
    }
    V3 h = reflect_half(wi, wo);
    float ndf = ggx_ndf(h.z, a2);
    float d = fabsf(wi.z) * kInvPi;
    float s = ndf / (4.0f * fabsf(dot(wi, h)));
    return dw * d + sw * s;
}
CPT_HD V3 micro_eval(const BsdfDev &b, V3 wo, V3 wi) {
    if (wo.z * wi.z <= 0.0f) {
        V3 h = refract_half(wi, wo, b.ior);
        float fr = fresnel(b.ior, wo, h);
        float ft = 1.0f - fr;
        float eta = wo.z >= 0.0f ? 1.0f / b.ior : b.ior;
        float a2 = b.roughness * b.roughness;
        float ndf = ggx_ndf(h.z, a2);
        float vis = smith_vis(fabsf(wo.z), fabsf(wi.z), a2);
        float denom = pow2(eta * dot(wo, h) + dot(wi, h));
        float num = 4.0f * fabsf(dot(wo, h)) * fabsf(dot(wi, h)) * eta * eta;
        return b.transmittance * ft * (1.0f - b.opacity) * ndf * vis * num / denom * fabsf(wi.z);
    }
    V3 h = reflect_half(wi, wo);
    float fr = fresnel(b.ior, wo, h);
    float ft = 1.0f - fr;
    float a2 = b.roughness * b.roughness;
    float ndf = ggx_ndf(h.z, a2);
    float vis = smith_vis(fabsf(wo.z), fabsf(wi.z), a2);
    V3 s = b.specular * fr * ndf * vis * fabsf(wi.z);
    V3 d = b.diffuse * ft * b.opacity * fabsf(wi.z) * kInvPi;
    return d + s;
}
CPT_HD BsdfSample micro_sample(const BsdfDev &b, V3 wo, float r1, V2 r2) {
    float sw, dw, tw;
    micro_weights(b, wo, sw, dw, tw);
    BsdfSample samp = empty_sample();
    if (r1 < dw) {
        samp.wi = cosine_hemisphere_sample(r2);
        samp.lobe = kLobeDiffuse;
    } else {
        float a2 = b.roughness * b.roughness;
        float phi = r2.x * k2Pi;
        float sin_phi = sinf(phi), cos_phi = cosf(phi);
        float cos_theta2 = (1.0f - r2.y) / (1.0f - r2.y * (1.0f - a2));
        float cos_theta = sqrtf(cos_theta2);
        float sin_theta = sqrtf(1.0f - cos_theta2);
        V3 h = V3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
        if (r1 < dw + sw) {
            samp.wi = reflect(wo, h);
            samp.lobe = kLobeGlossy;
        } else {
            samp.wi = refract(wo, h, b.ior);
            if (is_zero(samp.wi)) { samp.pdf = 0.0f; samp.lobe = kLobeNone; return samp; }
            samp.lobe = kLobeGlossy;
        }
    }
    samp.pdf = micro_pdf(b, wo, samp.wi);
    samp.weight = micro_eval(b, wo, samp.wi) / samp.pdf;
    return samp;
}

// ---- Glass (bsdf/glass.cuh:14-63): reflectance in b.diffuse, transmittance in b.specular
CPT_HD BsdfSample glass_sample(const BsdfDev &b, V3 wo, float r1) {
    float fr = fresnel(b.ior, wo, v3(0.0f, 0.0f, 1.0f));
    float ft = 1.0f - fr;
    float reflect_weight = fr * luminance(b.diffuse);
    float transmit_weight = ft * luminance(b.specular);
    float reflect_pdf = reflect_weight / (reflect_weight + transmit_weight);
    float transmit_pdf = 1.0f - reflect_pdf;
    BsdfSample samp = empty_sample();
    if (r1 < reflect_pdf) {
        samp.wi = reflect(wo, v3(0.0f, 0.0f, 1.0f));
        samp.pdf = reflect_pdf;
        samp.weight = fr * b.diffuse / reflect_pdf;
        samp.lobe = kLobeSpecular;
    } else {
        samp.wi = refract(wo, v3(0.0f, 0.0f, 1.0f), b.ior);
        if (is_zero(samp.wi)) {
            samp.pdf = 0.0f;
        } else {
            float eta = wo.z >= 0.0f ? 1.0f / b.ior : b.ior;
            samp.pdf = transmit_pdf;
            samp.weight = eta * eta * ft * b.specular / transmit_pdf;
            samp.lobe = kLobeSpecular;
        }
    }
    return samp;
}
CPT_HD float glass_pdf(const BsdfDev &b, V3 wo, V3 wi) {
    float fr = fresnel(b.ior, wo, v3(0.0f, 0.0f, 1.0f));
    float ft = 1.0f - fr;
    float reflect_weight = fr * luminance(b.diffuse);
    float transmit_weight = ft * luminance(b.specular);
    float reflect_pdf = reflect_weight / (reflect_weight + transmit_weight);
    float transmit_pdf = 1.0f - reflect_pdf;
    return wo.z * wi.z >= 0.0f ? reflect_pdf : transmit_pdf;
}
CPT_HD V3 glass_eval(const BsdfDev &b, V3 wo, V3 wi) {
    float fr = fresnel(b.ior, wo, v3(0.0f, 0.0f, 1.0f));
    float ft = 1.0f - fr;
    // both products are formed and one is chosen: a branch that returns a multiple of one member or of
    // another invites the compiler to pick the member by address, which parks the whole block in scratch
    float eta = wo.z >= 0.0f ? 1.0f / b.ior : b.ior;
    V3 transmitted = eta * eta * ft * b.specular, reflected = fr * b.diffuse;
    return wo.z * wi.z <= 0.0f ? transmitted : reflected;
}

// tagged dispatch, src/kernels/bsdf/bsdf.cuh:24-82
CPT_HD bool bsdf_is_delta(const BsdfDev &b) { return b.type == kGlass; }
CPT_HD BsdfSample bsdf_sample(const BsdfDev &b, V3 wo, float r1, V2 r2) {
    switch (b.type) {
        case kLambert: return lambert_sample(b, wo, r2);
        case kPhong: return phong_sample(b, wo, r1, r2);
        case kBlinnPhong: return blinn_sample(b, wo, r1, r2);
        case kMicrofacet: return micro_sample(b, wo, r1, r2);
        default: return glass_sample(b, wo, r1);
    }
}
CPT_HD float bsdf_pdf(const BsdfDev &b, V3 wo, V3 wi) {
    switch (b.type) {
        case kLambert: return lambert_pdf(wo, wi);
        case kPhong: return phong_pdf(b, wo, wi);
        case kBlinnPhong: return blinn_pdf(b, wo, wi);
        case kMicrofacet: return micro_pdf(b, wo, wi);
        default: return glass_pdf(b, wo, wi);
    }
}
CPT_HD V3 bsdf_eval(const BsdfDev &b, V3 wo, V3 wi) {
    switch (b.type) {
        case kLambert: return lambert_eval(b, wo, wi);
        case kPhong: return phong_eval(b, wo, wi);
        case kBlinnPhong: return blinn_eval(b, wo, wi);
        case kMicrofacet: return micro_eval(b, wo, wi);
        default: return glass_eval(b, wo, wi);
    }
}

// ------------------------------------------------------------------ textures
// What tex2D<float4> returns for the reference's texture objects (src/cuda_helpers/texture.cpp:
// normalised coordinates, wrap addressing, linear filtering, sRGB texels read as normalised
// floats): texel centres at (i + 0.5) / size, the two nearest texels per axis blended with the
// fractional distance kept to 8 bits (the fixed-point weights of the CUDA programming guide's
// texture-fetching appendix), sRGB decoded per texel before the blend.  The decode table is built
// once on the host so that every user of it sees the same 256 floats.
CPT_HD float wrap_unit(float x) { return x - floorf(x); }
CPT_D V3 texture_fetch(const TextureDev &t, const float *srgb_to_linear, float u, float v) {
    const float x = wrap_unit(u) * (float)t.width - 0.5f, y = wrap_unit(v) * (float)t.height - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f), b = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int w = (int)t.width, h = (int)t.height;
    int x0 = (int)fx % w, y0 = (int)fy % h;
    if (x0 < 0) x0 += w;
    if (y0 < 0) y0 += h;
    const int x1 = x0 + 1 == w ? 0 : x0 + 1, y1 = y0 + 1 == h ? 0 : y0 + 1;
    GlobalWords texels = (GlobalWords)t.texels;
    const uint32_t t00 = texels[y0 * w + x0], t10 = texels[y0 * w + x1], t01 = texels[y1 * w + x0], t11 = texels[y1 * w + x1];
    auto channel = [&](int shift) {
        const float c00 = srgb_to_linear[(t00 >> shift) & 255u], c10 = srgb_to_linear[(t10 >> shift) & 255u];
        const float c01 = srgb_to_linear[(t01 >> shift) & 255u], c11 = srgb_to_linear[(t11 >> shift) & 255u];
        return (1.0f - a) * (1.0f - b) * c00 + a * (1.0f - b) * c10 + (1.0f - a) * b * c01 + a * b * c11;
    };
    return V3{channel(0), channel(8), channel(16)};
}

// MaterialValue::At for every mapped value of a material (src/kernels/material/common.cuh:15-27:
// the map replaces the constant and is read at (u, 1 - v)), then the part of MtlMaterial::GetBsdf
// that depends on those values (src/kernels/material/mtl.cuh:29-58).
CPT_D BsdfDev material_at(const BsdfDev &m, const TextureDev *textures, const float *srgb_to_linear, V2 uv) {
    BsdfDev b = m;
    if ((m.tex_diffuse_specular | m.tex_emission) == 0u) return b;
    if (m.tex_emission) b.emission = texture_fetch(textures[m.tex_emission - 1], srgb_to_linear, uv.x, 1.0f - uv.y);
    const uint32_t td = m.tex_diffuse_specular & 0xffffu, ts = m.tex_diffuse_specular >> 16;
    if (b.type != kGlass && (td | ts) != 0u) {
        if (td) b.diffuse = texture_fetch(textures[td - 1], srgb_to_linear, uv.x, 1.0f - uv.y);
        if (ts) b.specular = texture_fetch(textures[ts - 1], srgb_to_linear, uv.x, 1.0f - uv.y);
        if (b.type == kPhong || b.type == kBlinnPhong) b.diffuse = b.diffuse * (1.0f - fmaxf(b.specular.x, fmaxf(b.specular.y, b.specular.z)));
    }
    return b;
}
CPT_HD bool material_reads_texcoord(const BsdfDev &m) { return (m.tex_diffuse_specular | m.tex_emission) != 0u; }

// Host-side evaluation of the MTL material into a BSDF block
// (src/kernels/material/material.cuh:15-27 and src/kernels/material/mtl.cuh:16-66).
// tex_diffuse / tex_specular / tex_emission: 1-based texture numbers or 0.  With a diffuse or
// specular map the Phong-family scaling of the diffuse colour is left to material_at().
inline BsdfDev make_bsdf(V3 emission, V3 diffuse, V3 specular, V3 transmittance, float shininess, float ior, float opacity, int bsdf_type,
                         uint32_t tex_diffuse = 0, uint32_t tex_specular = 0, uint32_t tex_emission = 0) {
    BsdfDev b;
    memset(&b, 0, sizeof(b));
    const bool mapped = (tex_diffuse | tex_specular) != 0;
    if (bsdf_type != kMicrofacet && opacity == 0.0f) {
        b.type = kGlass;
        b.specular = transmittance;
        b.diffuse = transmittance;      // reflectance := transmittance (mtl.cuh:20-22)
        b.ior = ior;
    } else {
        b.type = bsdf_type;
        switch (bsdf_type) {
            case kLambert: b.diffuse = diffuse; break;
            case kPhong:
                b.diffuse = diffuse; b.specular = specular;
                if (!mapped) b.diffuse = b.diffuse * (1.0f - fmaxf(b.specular.x, fmaxf(b.specular.y, b.specular.z)));
                b.shininess = shininess;
                b.s_norm = (shininess + 1) * kInv2Pi;
                break;
            case kBlinnPhong:
                b.diffuse = diffuse; b.specular = specular;
                if (!mapped) b.diffuse = b.diffuse * (1.0f - fmaxf(b.specular.x, fmaxf(b.specular.y, b.specular.z)));
                b.shininess = shininess;
                b.s_norm = (shininess + 2) / (4 * kPi * (2 - powf(2.0f, -shininess / 2)));
                break;
            case kMicrofacet:
                b.diffuse = diffuse; b.specular = specular; b.transmittance = transmittance;
                b.roughness = sqrtf(2.0f / (2.0f + shininess));
                b.ior = ior; b.opacity = opacity;
                break;
            default: break;
        }
    }
    b.emission = emission;
    b.tex_diffuse_specular = b.type == kGlass ? 0u : (tex_diffuse | (tex_specular << 16));
    b.tex_emission = tex_emission;
    return b;
}

// ------------------------------------------------------------------ camera (src/kernels/camera/pinhole.cuh:16-22)
CPT_HD void camera_sample_ray(const CameraDev &cam, float aspect, V2 pr, V3 &origin, V3 &dir) {
    V3 d = normalize((cam.x_dir * aspect * (pr.x - 0.5f)) + (cam.y_dir * (pr.y - 0.5f)) + v3(0.0f, 0.0f, -1.0f));
    Frame f{cam.frame_x, cam.frame_y, cam.frame_z};
    dir = f.to_world(d);
    origin = cam.pos;
}

// Camera ray of pixel (px, py) for 1-based sample index spp; leaves the sampler where the
// reference's RenderKernel leaves it before Trace (src/kernels/integrator/path.cu:86-96).
CPT_HD void pixel_ray(const CameraDev &cam, uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t spp, Sampler &s, uint32_t &pixel_index,
                      V3 &origin, V3 &dir) {
    pixel_index = (H - 1 - py) * W + px;
    s.state = rand_init_tea(pixel_index, spp);
    V2 sub = V2{0.5f, 0.5f};
    if (spp != 1) sub = s.next2d();
    V2 pr = V2{((float)px + sub.x) / (float)W, ((float)py + sub.y) / (float)H};
    camera_sample_ray(cam, (float)W / (float)H, pr, origin, dir);
    (void)s.next2d();   // aperture sample: drawn, unused
}

// integrator/common.cuh:9-25
CPT_HD float power_heuristic(float p0, float p1) { p0 *= p0; p1 *= p1; return p0 / (p0 + p1); }
CPT_HD float path_to_solid_angle_jacobian(V3 p_prev, V3 p, V3 n) {
    V3 vec = p_prev - p;
    float dist_sqr = dot(vec, vec);
    V3 dir = vec * inversesqrt(dist_sqr);
    float cos_theta = dot(n, dir);
    return cos_theta > 0.0f ? dist_sqr / cos_theta : 0.0f;
}

}  // namespace cpt
