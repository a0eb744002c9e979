#include "pathtracer.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../render.h"
#include "../shading.h"

using cpt::M4;
using cpt::V3;
using cpt::V4;

namespace {

inline void rc_check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString((hipError_t)rc));
}

// about 7 MB of node and triangle records: beyond that the BVH no longer lives in the caches next to the CUs
constexpr size_t kLargeSceneTriangles = 65536;
}  // namespace

PathTracer::PathTracer(Scene &scene, Film &film) : scene_(scene), film_(film) {
    film_zero_generation_ = film_.ZeroGeneration();
    if (film_.MappedSinceZeroed()) NoteFilmExposed();        // a film that has been in other hands: nothing is known about its alphas
}

// What the tracer believes about the film's alphas (all 0 or 1; all equal to a count of summed samples) holds for the
// content it left there.  A Film that was cleared or resized since is all zeros again, whoever did it.
void PathTracer::SyncFilmState() {
    if (film_.ZeroGeneration() == film_zero_generation_) return;
    film_zero_generation_ = film_.ZeroGeneration();
    film_alpha_unit_ = true;
    film_sum_count_ = 0;
}

PathTracer::~PathTracer() {
    workspace_.release();
    if (ev_begin_) (void)hipEventDestroy(ev_begin_);
    if (ev_end_) (void)hipEventDestroy(ev_end_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void PathTracer::BuildBuffers() {
    if (!stream_) CPT_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    BuildAccel();

    has_camera_ = CameraFromScene(scene_dev_.camera);

    BuildInstancesAndLights();
    size_t triangles = 0;
    scene_.ForEach<const MeshComponent>([&](const MeshComponent &mesh) { triangles += mesh.GetMesh()->IndicesCount() / 3; });
    scene_triangles_ = triangles;
    const bool dense = triangles >= kLargeSceneTriangles;
    const bool wide = scene_dev_.wide_nodes != 0;
    if (dense != wf_dense_ || wide != wf_wide_) { wf_dense_ = dense; wf_wide_ = wide; wf_.capacity = 0; }      // the resident workgroup counts are looked up again: they depend on the kernel build
    scene_dev_.large_scene = dense ? 1 : 0;
    CPT_HIP(hipStreamSynchronize(stream_));
    built_ = true;
}

// src/scene/camera.cpp:28-50 (CameraComponent::BuildBuffer); the last camera of the scene wins, as the reference's
// per-component buffers would leave it
bool PathTracer::CameraFromScene(cpt::CameraDev &out) {
    bool found = false;
    scene_.ForEach<CameraComponent>([&](CameraComponent &camera) {
        V3 frame_z = cpt::normalize(camera.pos - camera.look_at);
        V3 frame_x = cpt::normalize(cpt::cross(camera.up, frame_z));
        V3 frame_y = cpt::normalize(cpt::cross(frame_z, frame_x));
        float radians = (camera.fov * 0.5f) * 0.01745329251994329576923690768489f;
        float scale = std::tan(radians) * 2.0f;
        out = cpt::CameraDev{camera.pos, frame_x, frame_y, frame_z, V3{scale, 0.0f, 0.0f}, V3{0.0f, scale, 0.0f}};
        found = true;
    });
    return found;
}

// What the reference's panels do after an edit (src/scene/material.cpp:94-100: Material::SetChanged re-uploads the
// material's device block; src/scene/camera.cpp:19-25: CameraComponent::BuildBuffer re-uploads the camera) followed by
// PathTracer::ResetAccumelation.  Here the tracer looks for edits itself before it renders: a material whose
// revision moved gets its parameter block rebuilt, a camera whose values moved is re-derived, and either restarts
// the accumulation.  As in the reference, the light table is not remade (an edit does not turn an object into a
// light or back; BuildBuffers does).
bool PathTracer::RefreshEdits(bool restart) {
    bool changed = false;
    cpt::CameraDev camera = scene_dev_.camera;
    if (CameraFromScene(camera) && memcmp(&camera, &scene_dev_.camera, sizeof(camera)) != 0) {
        scene_dev_.camera = camera;
        changed = true;
    }
    bool material_moved = false;
    for (const MaterialEntry &m : material_entries_) material_moved |= m.material->revision != m.revision;
    if (material_moved) {
        std::vector<cpt::BsdfDev> blocks;
        for (MaterialEntry &m : material_entries_) {
            const Material *mat = m.material.get();
            blocks.push_back(cpt::make_bsdf(mat->emission, mat->diffuse, mat->specular, mat->transmittance, mat->shininess, mat->ior, mat->opacity, mat->bsdf_type,
                                            m.tex[0], m.tex[1], m.tex[2]));
            m.revision = mat->revision;
        }
        CPT_HIP(hipStreamSynchronize(stream_));          // no kernel reads the table while it is rewritten
        material_table_.SetData(blocks.data(), sizeof(cpt::BsdfDev) * blocks.size(), stream_);
        changed = true;
    }
    if (changed && !restart) restart_pending_ = true;      // the tables are current; the film is restarted by the next caller that owns it
    if ((changed || restart_pending_) && restart) {
        restart_pending_ = false;
        changed = true;
        ResetAccumelation();
        film_.Clear(stream_);
        film_alpha_unit_ = true;
        film_sum_count_ = 0;
    }
    return changed;
}

// src/pathtracer/pathtracer.cpp:91-166 and src/scene/mesh.cpp:84-149
void PathTracer::BuildAccel() {
    meshes_.clear();
    mesh_index_.clear();
    instance_mesh_.clear();
    memset(build_stage_ms_, 0, sizeof(build_stage_ms_));
    leaf_boxes_ms_ = pack_ms_ = 0.0f;
    largest_mesh_ = (size_t)-1;
    uint32_t largest = 0;

    const bool wide = node_records_ != 0;
    std::vector<cpt::Bbox> leaf_bboxes;
    scene_.ForEach<MeshComponent, const MaterialComponent>([&](SceneObject &object, MeshComponent &mesh_comp, const MaterialComponent &) {
        Mesh *mesh = mesh_comp.GetMesh();
        uint32_t n = (uint32_t)(mesh->IndicesCount() / 3);
        if (n == 0) return;
        if (mesh->Normals().size() != mesh->VericesCount() || mesh->Texcoords().size() != mesh->VericesCount())
            throw std::runtime_error("mesh '" + object.Name() + "' lacks per-vertex normals or texture coordinates");
        auto found = mesh_index_.find(mesh);
        if (found == mesh_index_.end()) {
            auto mb = std::make_unique<MeshBuffers>();
            mb->num_triangles = n;
            mb->positions.Allocate(sizeof(V3) * mesh->VericesCount(), mesh->Positions().data(), stream_);
            mb->normals.Allocate(sizeof(V3) * mesh->VericesCount(), mesh->Normals().data(), stream_);
            mb->texcoords.Allocate(sizeof(cpt::V2) * mesh->VericesCount(), mesh->Texcoords().data(), stream_);
            mb->indices.Allocate(sizeof(uint32_t) * mesh->IndicesCount(), mesh->Indices().data(), stream_);
            size_t num_nodes = 2 * (size_t)n - 1;
            mb->nodes.Allocate(sizeof(cpt::AccelNode) * num_nodes);
            mb->bboxes.Allocate(sizeof(cpt::Bbox) * num_nodes);
            mb->keys.Allocate(sizeof(uint64_t) * n);
            if (!wide) mb->packed_nodes.Allocate(sizeof(cpt::PackedNode) * (n > 1 ? n - 1 : 1));
            mb->tris.Allocate(sizeof(cpt::TriRecord) * n);
            leaf_scratch_.Allocate(sizeof(cpt::Bbox) * (size_t)n);

            // leaf boxes on the device (the reference loops over triangles on the host, mesh.cpp:88-98)
            const bool timed = profile_build_ && n >= largest;
            hipEvent_t box_ev[2] = {nullptr, nullptr};
            if (timed) {
                for (auto &e : box_ev) CPT_HIP(hipEventCreate(&e));
                CPT_HIP(hipEventRecord(box_ev[0], stream_));
            }
            rc_check(cpt::lbvh_triangle_boxes(leaf_scratch_.TypedGpuData<cpt::Bbox>(), mb->indices.TypedGpuData<uint32_t>(),
                                              mb->positions.TypedGpuData<V3>(), n, stream_), "triangle boxes");
            const struct Bbox &hb = mesh->Bbox();
            cpt::Bbox merged{V4{hb.pmin.x, hb.pmin.y, hb.pmin.z, 1.0f}, V4{hb.pmax.x, hb.pmax.y, hb.pmax.z, 1.0f}};
            if (timed) CPT_HIP(hipEventRecord(box_ev[1], stream_));
            bool profile = timed;
            float stage[5];
            mb->root.Allocate(sizeof(cpt::AccelRoot));
            if (wide) mb->wide_nodes.Allocate(sizeof(cpt::PackedNode) * 2 * (n > 1 ? n - 1 : 1));
            cpt::RecordOutputs records;
            records.binary = wide ? nullptr : mb->packed_nodes.TypedGpuData<cpt::PackedNode>();      // with two-level records the binary ones wait for a caller that walks them
            records.wide = wide ? mb->wide_nodes.TypedGpuData<cpt::PackedNode>() : nullptr;
            records.root = mb->root.TypedGpuData<cpt::AccelRoot>();
            rc_check(cpt::lbvh_build(mb->nodes.TypedGpuData<cpt::AccelNode>(), mb->bboxes.TypedGpuData<cpt::Bbox>(), merged, n, stream_, &workspace_,
                                     leaf_scratch_.TypedGpuData<cpt::Bbox>(), mb->keys.TypedGpuData<uint64_t>(), nullptr, profile ? stage : nullptr, &records),
                     "bottom-level build");
            if (profile) {
                memcpy(build_stage_ms_, stage, sizeof(stage));
                largest = n;
                largest_mesh_ = meshes_.size();
                CPT_HIP(hipEventSynchronize(box_ev[1]));
                CPT_HIP(hipEventElapsedTime(&leaf_boxes_ms_, box_ev[0], box_ev[1]));
            }
            for (auto &e : box_ev) if (e) (void)hipEventDestroy(e);
            found = mesh_index_.emplace(mesh, (uint32_t)meshes_.size()).first;
            meshes_.push_back(std::move(mb));
        }
        instance_mesh_.push_back(found->second);

        // top-level leaf = mesh bounds carried through the object's transform (pathtracer.cpp:103-109)
        struct Bbox bbox = mesh->Bbox().TransformBy(object.GetTransform());
        leaf_bboxes.push_back(cpt::Bbox{V4{bbox.pmin.x, bbox.pmin.y, bbox.pmin.z, 1.0f}, V4{bbox.pmax.x, bbox.pmax.y, bbox.pmax.z, 1.0f}});
    });

    uint32_t num_instances = (uint32_t)instance_mesh_.size();
    if (num_instances == 0) return;

    // A scene whose traversal records are a few KB gets them once more in one piece, for the kernel that keeps them in
    // LDS: the top level's node records, then per mesh its node records and its triangle records (16-byte words).
    const uint32_t top_record_count = num_instances > 1 ? num_instances - 1 : 0;
    uint32_t lds_words = top_record_count * cpt::kLdsNodeWords;
    std::vector<uint32_t> lds_nodes_at(meshes_.size()), lds_tris_at(meshes_.size());
    for (size_t i = 0; i < meshes_.size(); i++) {
        const uint32_t n = meshes_[i]->num_triangles;
        lds_nodes_at[i] = lds_words;
        if (lds_words <= cpt::kLdsSceneBytes / 16) lds_words += (n > 1 ? n - 1 : 0) * cpt::kLdsNodeWords;
        lds_tris_at[i] = lds_words;
        if (lds_words <= cpt::kLdsSceneBytes / 16) lds_words += 3 * n;
    }
    const bool lds_copy = wide && lds_words <= cpt::kLdsSceneBytes / 16;

    // mesh descriptor table; each root record is copied in from where its build left it
    std::vector<cpt::MeshDev> mesh_table(meshes_.size());
    for (size_t i = 0; i < meshes_.size(); i++) {
        MeshBuffers &mb = *meshes_[i];
        cpt::MeshDev &d = mesh_table[i];
        memset(&d, 0, sizeof(d));
        if (lds_copy) { d.lds_nodes = lds_nodes_at[i]; d.lds_tris = lds_tris_at[i]; }
        d.nodes = wide ? nullptr : mb.packed_nodes.TypedGpuData<cpt::PackedNode>();
        d.tris = mb.tris.TypedGpuData<cpt::TriRecord>();
        d.indices = mb.indices.TypedGpuData<uint32_t>();
        d.positions = mb.positions.TypedGpuData<V3>();
        d.normals = mb.normals.TypedGpuData<V3>();
        d.texcoords = mb.texcoords.TypedGpuData<cpt::V2>();
        d.num_triangles = mb.num_triangles;
        if (wide) d.wide = mb.wide_nodes.TypedGpuData<cpt::PackedNode>();
    }
    mesh_table_.Allocate(sizeof(cpt::MeshDev) * mesh_table.size(), mesh_table.data(), stream_);
    for (size_t i = 0; i < meshes_.size(); i++) {
        MeshBuffers &mb = *meshes_[i];
        cpt::AccelRoot *root = &mesh_table_.TypedGpuData<cpt::MeshDev>()[i].root;
        const bool timed = profile_build_ && i == largest_mesh_;
        hipEvent_t pack_ev[2] = {nullptr, nullptr};
        if (timed) {
            for (auto &e : pack_ev) CPT_HIP(hipEventCreate(&e));
            CPT_HIP(hipEventRecord(pack_ev[0], stream_));
        }
        struct PackTimer {               // closes the measurement when the iteration ends
            hipEvent_t *ev; hipStream_t stream; float *out; bool on;
            ~PackTimer() {
                if (!on) return;
                if (hipEventRecord(ev[1], stream) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess) (void)hipEventElapsedTime(out, ev[0], ev[1]);
                (void)hipEventDestroy(ev[0]);
                (void)hipEventDestroy(ev[1]);
            }
        } pack_timer{pack_ev, stream_, &pack_ms_, timed};
        CPT_HIP(hipMemcpyAsync(root, mb.root.GpuData(), sizeof(cpt::AccelRoot), hipMemcpyDeviceToDevice, stream_));
        rc_check(cpt::lbvh_pack_triangles(mb.tris.TypedGpuData<cpt::TriRecord>(), mb.nodes.TypedGpuData<cpt::AccelNode>(), mb.indices.TypedGpuData<uint32_t>(),
                                          mb.positions.TypedGpuData<V3>(), mb.num_triangles, stream_), "triangle records");
    }

    // top level (pathtracer.cpp:120-154)
    size_t num_nodes = 2 * (size_t)num_instances - 1;
    std::vector<cpt::Bbox> accel_bboxes(num_nodes);
    memset(accel_bboxes.data(), 0, sizeof(cpt::Bbox) * num_nodes);
    std::copy(leaf_bboxes.begin(), leaf_bboxes.end(), accel_bboxes.begin() + (num_instances - 1));
    const float fmax_ = 3.402823466e+38f;
    cpt::Bbox merged{V4{fmax_, fmax_, fmax_, 1.0f}, V4{-fmax_, -fmax_, -fmax_, 1.0f}};
    for (uint32_t i = 0; i < num_instances; i++) {
        const cpt::Bbox &b = accel_bboxes[num_instances - 1 + i];
        merged.pmin = V4{cpt::gmin(merged.pmin.x, b.pmin.x), cpt::gmin(merged.pmin.y, b.pmin.y), cpt::gmin(merged.pmin.z, b.pmin.z), cpt::gmin(merged.pmin.w, b.pmin.w)};
        merged.pmax = V4{cpt::gmax(merged.pmax.x, b.pmax.x), cpt::gmax(merged.pmax.y, b.pmax.y), cpt::gmax(merged.pmax.z, b.pmax.z), cpt::gmax(merged.pmax.w, b.pmax.w)};
    }
    top_bboxes_.Allocate(sizeof(cpt::Bbox) * num_nodes, accel_bboxes.data(), stream_);
    top_nodes_.Allocate(sizeof(cpt::AccelNode) * num_nodes);
    top_keys_.Allocate(sizeof(uint64_t) * num_instances);
    if (!wide) top_packed_.Allocate(sizeof(cpt::PackedNode) * (num_instances > 1 ? num_instances - 1 : 1));
    top_root_.Allocate(sizeof(cpt::AccelRoot));
    if (wide) top_wide_.Allocate(sizeof(cpt::PackedNode) * 2 * (num_instances > 1 ? num_instances - 1 : 1));
    cpt::RecordOutputs top_records;
    top_records.binary = wide ? nullptr : top_packed_.TypedGpuData<cpt::PackedNode>();
    top_records.wide = wide ? top_wide_.TypedGpuData<cpt::PackedNode>() : nullptr;
    top_records.root = top_root_.TypedGpuData<cpt::AccelRoot>();
    top_records.top_level = 1;
    rc_check(cpt::lbvh_build(top_nodes_.TypedGpuData<cpt::AccelNode>(), top_bboxes_.TypedGpuData<cpt::Bbox>(), merged, num_instances, stream_, &workspace_,
                             nullptr, top_keys_.TypedGpuData<uint64_t>(), nullptr, nullptr, &top_records), "top-level build");
    scene_dev_.top_wide = wide ? top_wide_.TypedGpuData<cpt::PackedNode>() : nullptr;
    scene_dev_.wide_nodes = wide ? 1 : 0;
    scene_dev_.top_nodes = wide ? nullptr : top_packed_.TypedGpuData<cpt::PackedNode>();
    binary_records_ = !wide;
    scene_dev_.top_root = top_root_.TypedGpuData<cpt::AccelRoot>();
    scene_dev_.meshes = mesh_table_.TypedGpuData<cpt::MeshDev>();

    scene_dev_.lds_scene = nullptr;
    scene_dev_.lds_scene_words = 0;
    scene_dev_.lds_top = 0;
    lds_scene_.Release();
    if (lds_copy && lds_words != 0) {
        lds_scene_.Allocate(16 * (size_t)lds_words);
        float *blob = lds_scene_.TypedGpuData<float>();
        rc_check(cpt::launch_lds_scene_nodes(blob, top_wide_.TypedGpuData<cpt::PackedNode>(), top_record_count, stream_), "records for LDS, top level");
        for (size_t i = 0; i < meshes_.size(); i++) {
            MeshBuffers &mb = *meshes_[i];
            const uint32_t n = mb.num_triangles;
            rc_check(cpt::launch_lds_scene_nodes(blob + 4 * (size_t)lds_nodes_at[i], mb.wide_nodes.TypedGpuData<cpt::PackedNode>(), n > 1 ? n - 1 : 0, stream_), "records for LDS");
            CPT_HIP(hipMemcpyAsync(blob + 4 * (size_t)lds_tris_at[i], mb.tris.GpuData(), sizeof(cpt::TriRecord) * (size_t)n, hipMemcpyDeviceToDevice, stream_));
        }
        scene_dev_.lds_scene = blob;
        scene_dev_.lds_scene_words = lds_words;
    }
}

// The binary node records, for the ordered batch queries: scenes built with two-level records get them the first
// time such a query asks (0.64 GB of stores per 10M triangles that most renders never need).
void PathTracer::EnsureBinaryRecords() {
    if (binary_records_ || !built_ || instance_mesh_.empty()) return;
    std::vector<const cpt::PackedNode *> table_entries(meshes_.size());      // lives until the copies below have been waited for
    for (size_t i = 0; i < meshes_.size(); i++) {
        MeshBuffers &mb = *meshes_[i];
        const uint32_t n = mb.num_triangles;
        mb.packed_nodes.Allocate(sizeof(cpt::PackedNode) * (n > 1 ? n - 1 : 1));
        rc_check(cpt::lbvh_pack_binary(mb.packed_nodes.TypedGpuData<cpt::PackedNode>(), mb.nodes.TypedGpuData<cpt::AccelNode>(), mb.bboxes.TypedGpuData<cpt::Bbox>(), n, false, stream_),
                 "binary records");
        table_entries[i] = mb.packed_nodes.TypedGpuData<cpt::PackedNode>();
        CPT_HIP(hipMemcpyAsync(&mesh_table_.TypedGpuData<cpt::MeshDev>()[i].nodes, &table_entries[i], sizeof(table_entries[i]), hipMemcpyHostToDevice, stream_));
    }
    const uint32_t num_instances = (uint32_t)instance_mesh_.size();
    top_packed_.Allocate(sizeof(cpt::PackedNode) * (num_instances > 1 ? num_instances - 1 : 1));
    rc_check(cpt::lbvh_pack_binary(top_packed_.TypedGpuData<cpt::PackedNode>(), top_nodes_.TypedGpuData<cpt::AccelNode>(), top_bboxes_.TypedGpuData<cpt::Bbox>(), num_instances, true, stream_),
             "binary records, top level");
    scene_dev_.top_nodes = top_packed_.TypedGpuData<cpt::PackedNode>();
    CPT_HIP(hipStreamSynchronize(stream_));
    binary_records_ = true;
}

// src/pathtracer/pathtracer.cpp:168-221
void PathTracer::BuildInstancesAndLights() {
    std::vector<cpt::InstanceDev> instances;
    std::vector<cpt::BsdfDev> materials;
    std::vector<uint32_t> lights;
    std::unordered_map<const Material *, uint32_t> material_index;
    material_entries_.clear();
    uint32_t slot = 0;
    // one device texture per distinct image, numbered from 1
    std::vector<const Image *> images;
    std::unordered_map<const Image *, uint32_t> image_index;
    auto texture_number = [&](const std::shared_ptr<Image> &image) -> uint32_t {
        if (!image || image->rgba.empty()) return 0u;
        auto found = image_index.find(image.get());
        if (found == image_index.end()) {
            images.push_back(image.get());
            found = image_index.emplace(image.get(), (uint32_t)images.size()).first;
        }
        return found->second;
    };

    scene_.ForEach<const MeshComponent, const MaterialComponent>([&](SceneObject &object, const MeshComponent &mesh, const MaterialComponent &material) {
        if (mesh.GetMesh()->IndicesCount() / 3 == 0) return;
        const Material *mat = material.GetMaterial();
        auto found = material_index.find(mat);
        if (found == material_index.end()) {
            found = material_index.emplace(mat, (uint32_t)materials.size()).first;
            MaterialEntry entry{material.Shared(), mat->revision, {texture_number(mat->diffuse_image), texture_number(mat->specular_image), texture_number(mat->emission_image)}};
            materials.push_back(cpt::make_bsdf(mat->emission, mat->diffuse, mat->specular, mat->transmittance, mat->shininess, mat->ior, mat->opacity,
                                               mat->bsdf_type, entry.tex[0], entry.tex[1], entry.tex[2]));
            material_entries_.push_back(std::move(entry));
        }
        cpt::InstanceDev inst;
        memset(&inst, 0, sizeof(inst));
        inst.transform = object.GetTransform();
        inst.transform_inv = InverseTransform(inst.transform);
        inst.transform_it = cpt::transpose(inst.transform_inv);
        {
            const float *m = &inst.transform_inv.c[0].x;      // m[4 * column + row]
            bool identity = true;
            for (int k = 0; k < 16; k++) identity = identity && (k % 5 == 0 ? m[k] == 1.0f : m[k] == 0.0f);
            // component r of the product: (m[r]*x + m[4+r]*y) + (m[8+r]*z + m[12+r]*1); the zeros next to the 1 of row r, in term order
            for (int r = 0; r < 3; r++) {
                uint32_t word = 0, bit = 0x80000000u;
                for (int c = 0; c < 4; c++) {
                    if (c == r) continue;
                    if (std::signbit(m[4 * c + r])) word |= bit;
                    bit >>= 1;
                }
                inst.zero_sign[r] = word;
            }
            if (identity) inst.flags |= cpt::kInstanceIdentity;
        }
        inst.mesh = instance_mesh_[slot];
        inst.material = found->second;
        inst.light = -1;
        if (mat->IsEmissive()) {
            inst.light = (int32_t)lights.size();
            lights.push_back(slot);
        }
        instances.push_back(inst);
        slot++;
    });

    // position of every instance among the sorted top-level leaves: the low word of sorted key k is the instance at position k
    if (!instances.empty()) {
        std::vector<uint64_t> sorted(instances.size());
        CPT_HIP(hipMemcpyAsync(sorted.data(), top_keys_.GpuData(), sizeof(uint64_t) * sorted.size(), hipMemcpyDeviceToHost, stream_));
        CPT_HIP(hipStreamSynchronize(stream_));
        for (size_t k = 0; k < sorted.size(); k++) {
            const uint32_t id = (uint32_t)sorted[k];
            if (id < instances.size()) instances[id].tlas_position = (uint32_t)k;
        }
    }
    num_lights_ = (uint32_t)lights.size();
    if (!instances.empty()) instance_table_.Allocate(sizeof(cpt::InstanceDev) * instances.size(), instances.data(), stream_);
    if (!materials.empty()) material_table_.Allocate(sizeof(cpt::BsdfDev) * materials.size(), materials.data(), stream_);
    if (!lights.empty()) light_table_.Allocate(sizeof(uint32_t) * lights.size(), lights.data(), stream_);
    counters_.Allocate(sizeof(cpt::Counters) * cpt::kCounterSlots);
    counters_.Zero(stream_);
    if (images.size() > 0xffffu) throw std::runtime_error("more than 65535 distinct material maps");
    scene_dev_.textures = nullptr;
    scene_dev_.srgb_to_linear = nullptr;
    if (!images.empty()) {
        std::vector<uint32_t> all;
        std::vector<size_t> first;
        for (const Image *im : images) {
            first.push_back(all.size());
            all.insert(all.end(), im->rgba.begin(), im->rgba.end());
        }
        texel_pool_.Allocate(all.size() * sizeof(uint32_t), all.data(), stream_);
        const uint32_t *pool = texel_pool_.TypedGpuData<uint32_t>();
        std::vector<cpt::TextureDev> table;
        for (size_t k = 0; k < images.size(); k++) table.push_back(cpt::TextureDev{pool + first[k], images[k]->width, images[k]->height});
        texture_table_.Allocate(sizeof(cpt::TextureDev) * table.size(), table.data(), stream_);
        // sRGB electro-optical transfer function (IEC 61966-2-1) for the 256 codes
        float lut[256];
        for (int c = 0; c < 256; c++) {
            const float s = (float)c / 255.0f;
            lut[c] = s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f);
        }
        srgb_table_.Allocate(sizeof(lut), lut, stream_);
        scene_dev_.textures = texture_table_.TypedGpuData<cpt::TextureDev>();
        scene_dev_.srgb_to_linear = srgb_table_.TypedGpuData<float>();
    }

    scene_dev_.instances = instance_table_.TypedGpuData<cpt::InstanceDev>();
    scene_dev_.materials = material_table_.TypedGpuData<cpt::BsdfDev>();
    scene_dev_.lights = light_table_.TypedGpuData<uint32_t>();
    scene_dev_.num_lights = num_lights_;
    scene_dev_.light_pick_pdf = num_lights_ != 0 ? 1.0f / num_lights_ : 0.0f;    // pathtracer.cpp:47
    scene_dev_.counters = counters_.TypedGpuData<cpt::Counters>();
}

void PathTracer::ResetAccumelation() { curr_spp_ = 0; }

// src/pathtracer/pathtracer.cpp:26-61
void PathTracer::Update() {
    if (!built_ || !has_camera_ || num_lights_ == 0 || instance_mesh_.empty()) return;
    RefreshEdits();
    if (film_.Width() != last_width_ || film_.Height() != last_height_) {
        ResetAccumelation();
        if (last_width_ != 0) { film_alpha_unit_ = true; film_sum_count_ = 0; }       // a resized film starts from zeros
        last_width_ = film_.Width();
        last_height_ = film_.Height();
    }
    ++curr_spp_;
    cpt::RenderParams p;
    p.output = film_.DeviceMap(stream_);
    p.width = film_.Width();
    p.height = film_.Height();
    p.spp = curr_spp_;
    p.max_depth = static_cast<uint32_t>(max_depth_);
    p.channel = display_channel_;
    p.sum_mode = 0;
    SyncFilmState();
    LaunchSample(p);
    film_sum_count_ = -1;            // a running mean leaves another alpha
    film_.DeviceUnmap();
}

// `count` times Update(), with the launches of several samples folded into one where the kernel in use can do
// that: a launch's start and the tail of its longest paths cost about as much as a third of a 1024 x 1024 sample.
void PathTracer::UpdateMany(uint32_t count) {
    if (!built_ || !has_camera_ || num_lights_ == 0 || instance_mesh_.empty() || count == 0) return;
    RefreshEdits();
    if (film_.Width() != last_width_ || film_.Height() != last_height_) {
        ResetAccumelation();
        if (last_width_ != 0) { film_alpha_unit_ = true; film_sum_count_ = 0; }       // a resized film starts from zeros
        last_width_ = film_.Width();
        last_height_ = film_.Height();
    }
    cpt::RenderParams p;
    p.output = film_.DeviceMap(stream_);
    p.width = film_.Width();
    p.height = film_.Height();
    p.spp = curr_spp_ + 1;
    p.max_depth = static_cast<uint32_t>(max_depth_);
    p.channel = display_channel_;
    p.sum_mode = 0;
    LaunchSamples(p, count);
    curr_spp_ += count;
    film_.DeviceUnmap();
}

void PathTracer::LaunchSamples(cpt::RenderParams p, uint32_t count) {
    SyncFilmState();
    const int integrator = integrator_ == 3 ? (wf_dense_ ? 4 : 5) : integrator_;
    const bool folds = (integrator == 4 || integrator == 5) && samples_per_launch_ > 1;
    while (count != 0) {
        uint32_t batch = 1;
        if (folds && count > 1) {
            const unsigned long long tiles = cpt::render_tile_count(p.width, p.height);
            const unsigned long long fits = tiles ? 0xffffffffull / (tiles * cpt::kSegmentSlots) : 1;       // slot numbers are 32 bits
            batch = (uint32_t)std::min<unsigned long long>({(unsigned long long)count, (unsigned long long)samples_per_launch_, fits ? fits : 1});
        }
        p.turn_in_alpha = 0;
        p.alpha_base = 0;
        const bool film_memory = p.output == film_.DeviceMap(stream_);
        const bool own_film = film_memory && p.width == film_.Width() && p.height == film_.Height();
        if (film_memory && !own_film) { film_alpha_unit_ = false; film_sum_count_ = -1; }       // a window of other extents over the film's memory
        if (batch > 1 && !p.sum_mode && film_alpha_unit_ && own_film) {
            p.turn_in_alpha = 1;           // the pixel's alpha carries the count during the launch
            p.order = nullptr;
        } else if (batch > 1 && p.sum_mode && own_film && film_sum_count_ >= 0 && film_sum_count_ + batch < (1ll << 24)) {
            p.turn_in_alpha = 1;           // sums: the alpha is the count
            p.alpha_base = (uint32_t)film_sum_count_;
            p.order = nullptr;
        } else if (batch > 1) {
            const size_t words = (size_t)p.width * p.height;
            if (order_.Size() != words * sizeof(uint32_t)) order_.Allocate(words * sizeof(uint32_t));
            order_.Zero(stream_);
            p.order = order_.TypedGpuData<uint32_t>();
        } else {
            p.order = nullptr;
        }
        p.spp_count = batch;
        LaunchSample(p);
        if (own_film) {
            if (!p.sum_mode) film_sum_count_ = -1;                   // running means leave another alpha
            else if (film_sum_count_ >= 0) film_sum_count_ += batch;
        }
        p.spp += batch;
        count -= batch;
    }
}

void PathTracer::LaunchWavefront(const cpt::RenderParams &p) {
    size_t slots = cpt::wavefront_slots(p.width, p.height);
    if (wf_.capacity != slots) {
        size_t segments = slots / cpt::kSegmentSlots;
        wf_state_.Allocate(slots * 8 * sizeof(float4));
        wf_queues_.Allocate((slots * 2 + cpt::kControlWords) * sizeof(uint32_t));
        float4 *s = wf_state_.TypedGpuData<float4>();
        wf_.ray_o = s; wf_.ray_d = s + slots; wf_.thr = s + 2 * slots; wf_.col = s + 3 * slots;
        wf_.hit = s + 4 * slots; wf_.sh_o = s + 5 * slots; wf_.sh_d = s + 6 * slots; wf_.pending = s + 7 * slots;
        uint32_t *q = wf_queues_.TypedGpuData<uint32_t>();
        wf_.control = q;
        wf_.queue = q + cpt::kControlWords;
        wf_.shadow_queue = wf_.queue + slots;
        wf_.capacity = (uint32_t)slots;
        wf_.segments = (uint32_t)segments;
        int device = 0;
        CPT_HIP(hipGetDevice(&device));
        wf_resident_blocks_ = cpt::wavefront_resident_blocks(device, wf_dense_, scene_dev_.wide_nodes != 0);
        wf_finish_blocks_ = cpt::wavefront_finish_resident_blocks(device, scene_dev_.wide_nodes != 0);
    }
    const uint32_t resident = (uint32_t)wf_resident_blocks_;
    rc_check(cpt::wavefront_begin_sample(scene_dev_, p, wf_, stream_), "wavefront generate");
    // a path asks for at most max_depth + 1 closest-hit rays (camera ray included)
    const bool bounded = p.max_depth != 0xffffffffu;
    const uint32_t needed = bounded ? p.max_depth + 1 : 0xffffffffu;
    const bool pure = wavefront_tail_ < 0;
    uint32_t queued = needed;
    if (!pure && (uint32_t)wavefront_tail_ < queued) queued = (uint32_t)wavefront_tail_;
    if (queued != 0xffffffffu) {
        for (uint32_t b = 0; b < queued; b++) rc_check(cpt::wavefront_bounce(scene_dev_, p, wf_, b + 1, resident, wf_dense_, stream_), "wavefront bounce");
        if (queued < needed) rc_check(cpt::wavefront_finish(scene_dev_, p, wf_, (uint32_t)wf_finish_blocks_, stream_), "wavefront finish");
        return;
    }
    // queues only, unbounded depth: Russian roulette ends every path; the shade kernel stores the
    // bounce number while any path goes on, read it back after each group of bounces
    uint32_t bounce = 0, group = 4;
    for (;;) {
        for (uint32_t k = 0; k < group; k++) rc_check(cpt::wavefront_bounce(scene_dev_, p, wf_, ++bounce, resident, wf_dense_, stream_), "wavefront bounce");
        uint32_t marker = 0;
        CPT_HIP(hipMemcpyAsync(&marker, wf_.control + cpt::kControlLiveMarker, sizeof(marker), hipMemcpyDeviceToHost, stream_));
        CPT_HIP(hipStreamSynchronize(stream_));
        if (marker != bounce || bounce >= (1u << 20)) break;
        if (group < 64) group *= 2;
    }
}

void PathTracer::LaunchSample(const cpt::RenderParams &p) {
    // 3 = by scene size: the persistent kernel that shades partly finished waves (4) is ahead of every other once
    // rays are long (measured at 1M and 10M triangles); on a scene as small as the Cornell box the one that shades
    // whole waves and lets finished lanes walk the pooled shadow rays (5) is
    const int integrator = integrator_ == 3 ? (wf_dense_ ? 4 : 5) : integrator_;
    if (integrator == 2) {
        LaunchWavefront(p);
        return;
    }
    if (integrator == 1) {
        rc_check(cpt::launch_render(scene_dev_, p, stream_), "render");
        return;
    }
    const int variant = integrator == 4 ? 1 : (integrator == 5 ? 2 : 0);
    const int fast = (scene_dev_.fast_traversal != 0 ? 1 : 0) | (scene_dev_.wide_nodes != 0 ? 2 : 0);      // traversal of the kernel build
    // the copy of the records that the small-scene kernel keeps in LDS goes with the launch only when it is wanted
    cpt::SceneDev scene = scene_dev_;
    if (!scene_in_lds_) scene.lds_scene = nullptr;
    const bool in_lds = variant == 2 && scene.lds_scene != nullptr && !wf_dense_ && fast == 2;
    if (!queue_.Size() || resident_dense_ != wf_dense_ || resident_variant_ != variant || resident_traversal_ != fast || resident_scene_in_lds_ != in_lds) {
        queue_.Allocate(cpt::kRegenQueueWords * sizeof(uint32_t));
        queue_.Zero(stream_);
        queue_parity_ = 0;
        int device = 0;
        CPT_HIP(hipGetDevice(&device));
        resident_blocks_ = cpt::render_regen_resident_blocks(device, wf_dense_, variant, (fast & 1) != 0, (fast & 2) != 0, in_lds);
        resident_scene_in_lds_ = in_lds;
        resident_dense_ = wf_dense_;
        resident_variant_ = variant;
        resident_traversal_ = fast;
    }
    rc_check(cpt::launch_render_regen(scene, p, queue_.TypedGpuData<uint32_t>(), queue_parity_, (uint32_t)resident_blocks_, wf_dense_, variant, stream_), "render");
    queue_parity_ ^= 1u;
}

void PathTracer::RenderSamples(uint32_t spp_begin, uint32_t count, bool sum_mode, float *output, uint32_t width, uint32_t height) {
    if (!built_ || !has_camera_ || num_lights_ == 0 || instance_mesh_.empty()) return;
    RefreshEdits();
    SyncFilmState();
    // the film's alpha becomes a sample count, also when the caller passes the film's own pointer back in
    if (sum_mode && (!output || output == film_.DeviceMap(stream_))) film_alpha_unit_ = false;
    cpt::RenderParams p;
    p.output = output ? output : film_.DeviceMap(stream_);
    p.width = output ? width : film_.Width();
    p.height = output ? height : film_.Height();
    p.max_depth = static_cast<uint32_t>(max_depth_);
    p.channel = display_channel_;
    p.sum_mode = sum_mode ? 1 : 0;
    p.spp = spp_begin;
    LaunchSamples(p, count);
}

// kernel::PathTracer::Render(const Params &) (src/kernels/integrator/path.cu:109-115): one sample per pixel with the
// values of the call, into `output`, finished on return; the tracer's own settings, film and sample counter are not
// touched (edited materials and a moved camera reach the device tables; the restart of the tracer's own accumulation
// that such an edit asks for is left to the next Update).  A scene without lights renders nothing, as
// PathTracer::Update returns early for it (src/pathtracer/pathtracer.cpp:27-29): the light table does not exist.
void PathTracer::RenderOneSample(float *output, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth, int channel) {
    if (!built_ || !has_camera_ || instance_mesh_.empty()) throw std::runtime_error("PathTracer::RenderOneSample before BuildBuffers produced a scene");
    if (!output || width == 0 || height == 0 || spp == 0) throw std::invalid_argument("PathTracer::RenderOneSample: bad argument");
    if (num_lights_ == 0) throw std::invalid_argument("PathTracer::RenderOneSample: the scene has no emissive object");
    RefreshEdits(false);
    cpt::RenderParams p;
    p.output = output;
    p.width = width;
    p.height = height;
    p.spp = spp;
    p.max_depth = max_depth;
    p.channel = channel;
    p.sum_mode = 0;
    LaunchSample(p);
    CPT_HIP(hipStreamSynchronize(stream_));
}

void PathTracer::FirstHit(const float *d_rays, uint64_t num_rays, uint32_t *d_instance, uint32_t *d_primitive, float *d_t, float *d_uv) {
    const cpt::SceneDev &sc = DeviceSceneForQueries();
    if (!query_stream_ || sc.fast_traversal != 0 || num_rays < kStreamQueryRays || num_rays >= 0xffffffc0ull) {
        rc_check(cpt::launch_first_hit(sc, d_rays, num_rays, d_instance, d_primitive, d_t, d_uv, stream_), "first_hit");
        return;
    }
    if (!query_queue_.Size()) {
        query_queue_.Allocate(cpt::kRegenQueueWords * sizeof(uint32_t));
        query_queue_.Zero(stream_);
        query_parity_ = 0;
    }
    const bool fast = sc.fast_traversal != 0, wide = !fast && sc.wide_nodes != 0, large = sc.large_scene != 0;
    const int key = (large ? 1 : 0) | (fast ? 2 : 0) | (wide ? 4 : 0);
    if (key != query_key_) {
        int device = 0;
        CPT_HIP(hipGetDevice(&device));
        query_blocks_ = cpt::first_hit_stream_resident_blocks(device, large, fast, wide);
        query_key_ = key;
    }
    rc_check(cpt::launch_first_hit_stream(sc, d_rays, num_rays, d_instance, d_primitive, d_t, d_uv, query_queue_.TypedGpuData<uint32_t>(), query_parity_,
                                          (uint32_t)query_blocks_, stream_), "first_hit");
    query_parity_ ^= 1u;
}

void PathTracer::Synchronize() {
    if (stream_) CPT_HIP(hipStreamSynchronize(stream_));
}

void PathTracer::TimerBegin() {
    if (!stream_) return;
    if (!ev_begin_) { CPT_HIP(hipEventCreate(&ev_begin_)); CPT_HIP(hipEventCreate(&ev_end_)); }
    CPT_HIP(hipEventRecord(ev_begin_, stream_));
}

float PathTracer::TimerEnd() {
    if (!stream_ || !ev_begin_) return 0.0f;
    CPT_HIP(hipEventRecord(ev_end_, stream_));
    CPT_HIP(hipEventSynchronize(ev_end_));
    float ms = 0.0f;
    CPT_HIP(hipEventElapsedTime(&ms, ev_begin_, ev_end_));
    return ms;
}

float PathTracer::UpdateTimed(uint32_t count) {
    if (!stream_) return 0.0f;
    if (!ev_begin_) { CPT_HIP(hipEventCreate(&ev_begin_)); CPT_HIP(hipEventCreate(&ev_end_)); }
    CPT_HIP(hipEventRecord(ev_begin_, stream_));
    UpdateMany(count);
    CPT_HIP(hipEventRecord(ev_end_, stream_));
    CPT_HIP(hipEventSynchronize(ev_end_));
    float ms = 0.0f;
    CPT_HIP(hipEventElapsedTime(&ms, ev_begin_, ev_end_));
    return ms;
}

void PathTracer::ReadCounters(unsigned long long out[6]) {
    for (int i = 0; i < 6; i++) out[i] = 0;
    if (!counters_.Size()) return;
    std::vector<cpt::Counters> slots(cpt::kCounterSlots);
    CPT_HIP(hipMemcpyAsync(slots.data(), counters_.GpuData(), sizeof(cpt::Counters) * cpt::kCounterSlots, hipMemcpyDeviceToHost, stream_));
    CPT_HIP(hipStreamSynchronize(stream_));
    for (const cpt::Counters &c : slots) {
        out[0] += c.rays; out[1] += c.node_visits; out[2] += c.tri_tests; out[3] += c.paths; out[4] += c.shades; out[5] += c.stack_overflows;
    }
}

void PathTracer::ResetCounters() {
    if (counters_.Size()) counters_.Zero(stream_);
}

PathTracer::AccelView PathTracer::Accel(int mesh) const {
    if (mesh < 0)
        return AccelView{top_nodes_.TypedGpuData<cpt::AccelNode>(), top_bboxes_.TypedGpuData<cpt::Bbox>(), top_keys_.TypedGpuData<uint64_t>(),
                         (uint32_t)instance_mesh_.size()};
    if ((size_t)mesh >= instance_mesh_.size()) return AccelView{nullptr, nullptr, nullptr, 0};
    const MeshBuffers &mb = *meshes_[instance_mesh_[mesh]];
    return AccelView{mb.nodes.TypedGpuData<cpt::AccelNode>(), mb.bboxes.TypedGpuData<cpt::Bbox>(), mb.keys.TypedGpuData<uint64_t>(), mb.num_triangles};
}
