// Host side of the renderer: the reference's PathTracer component
// (src/pathtracer/pathtracer.hpp:7-50) on the HIP runtime.  BuildBuffers turns the host scene
// into device tables (per-mesh bottom-level BVH, top-level BVH over instances, instance / material
// / light tables, camera); Update renders one more sample per pixel into the Film.
#pragma once
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../device_types.h"
#include "../lbvh.h"
#include "../wavefront.h"
#include "buffer.hpp"
#include "film.hpp"
#include "scene.hpp"

class PathTracer {
public:
    PathTracer(Scene &scene, Film &film);
    ~PathTracer();

    void BuildBuffers();
    void ResetAccumelation();        // [sic] the reference's spelling
    void Update();
    // `count` more samples per pixel: the film ends up bit for bit as after `count` Update() calls, but a kernel that
    // can render several samples per launch is launched once per SetSamplesPerLaunch() samples
    void UpdateMany(uint32_t count);
    void SetSamplesPerLaunch(uint32_t samples) { samples_per_launch_ = samples < 1 ? 1 : samples; }
    void ShowUi() {}                 // the interactive panel is out of scope
    void SetCaptureName(std::string_view capture_name) { capture_name_ = capture_name; }
    void SetMaxDepth(int max_depth) { max_depth_ = max_depth; }

    // ---- additions for batch use (tests, benchmark, multi-GPU sharding)
    void SetChannel(int channel) { display_channel_ = channel; }
    // the film's memory was cleared / may have been written by someone else (its device pointer was handed out)
    // (a clear or resize of the Film itself is noticed without being told: SyncFilmState)
    void NoteFilmCleared() { film_alpha_unit_ = true; film_sum_count_ = 0; }
    void NoteFilmExposed() { film_alpha_unit_ = false; film_sum_count_ = -1; }
    // 0: the reference's traversal order and arithmetic, results bit for bit the reference's (default); 1: ordered
    // traversal (nearer child first, slab distances by reciprocal multiplication): same hit distances, but among
    // triangles at (nearly) the same distance another one may be reported.  Used by the batch queries and by
    // integrators 4 and 5 (and 3, which picks one of them); integrators 0, 1 and 2 always run exact.
    void SetTraversal(int mode) { scene_dev_.fast_traversal = mode ? 1 : 0; }
    int Traversal() const { return scene_dev_.fast_traversal; }
    // Two-level node records next to the binary ones (128 bytes per internal node: the node's grandchildren in the order
    // the reference reaches them), walked by the exact traversal of every kernel: half as many dependent record fetches
    // and traversal trips per ray, the same leaves in the same order.  1: built (default); 0: binary records only.
    // Takes effect at the next BuildBuffers.
    void SetNodeRecords(int mode) { node_records_ = mode; }
    bool HasTwoLevelRecords() const { return scene_dev_.wide_nodes != 0; }
    // 0: persistent lane-regenerating megakernel (default); 1: one path per lane, one lane per pixel;
    // 2: wavefront (ray queues in HBM, one kernel sequence per bounce); 4: persistent megakernel that keeps traversal
    // state across shading; 5: persistent megakernel whose finished lanes walk pooled shadow rays; 3: 4 or 5 by the
    // number of triangles (4 from 65,536 on)
    void SetIntegrator(int mode) { integrator_ = mode; }
    // wavefront only: queue bounces before the persistent finishing launch; negative = queues only
    void SetWavefrontTail(int bounces) { wavefront_tail_ = bounces; }
    void Synchronize();
    uint32_t CurrentSpp() const { return curr_spp_; }
    uint32_t NumLights() const { return num_lights_; }
    uint32_t NumInstances() const { return (uint32_t)instance_mesh_.size(); }
    hipStream_t Stream() const { return stream_; }
    const cpt::SceneDev &DeviceScene() const { return scene_dev_; }
    // the same for a batch query (first hit, any hit): the ordered traversal of those walks the binary node records,
    // which a scene built with two-level records gets when the first such query arrives
    const cpt::SceneDev &DeviceSceneForQueries() { if (scene_dev_.fast_traversal) EnsureBinaryRecords(); return scene_dev_; }
    // AccelTop::Intersect for a batch of rays (src/kernels/accel/accel.cuh:124-150) on the tracer's stream, asynchronous.
    // SetQueryKernel(1): batches of kStreamQueryRays rays and more, in exact traversal, run a persistent kernel whose lanes
    // take the next ray when theirs is finished: a quarter to a third faster on rays that diverge, slower on coherent ones
    // (camera rays), so the default (0) is one ray per lane for the whole launch.  Same results either way.
    void FirstHit(const float *d_rays, uint64_t num_rays, uint32_t *d_instance, uint32_t *d_primitive, float *d_t, float *d_uv);
    void SetQueryKernel(int mode) { query_stream_ = mode != 0; }
    // Whether the kernel for small scenes reads node and triangle records from a copy in LDS (default) or from memory
    // like every other kernel: same results.  Scenes whose records exceed cpt::kLdsSceneBytes have no such copy.
    void SetSceneInLds(bool on) { scene_in_lds_ = on; }
    bool SceneInLds() const { return scene_in_lds_ && lds_scene_.Size() != 0; }
    static constexpr uint64_t kStreamQueryRays = 1u << 16;
    // Renders sample indices [spp_begin, spp_begin + count) (1-based) into `output` (device,
    // W*H float4; the Film when null).  sum_mode adds radiance instead of keeping a running mean.
    void RenderSamples(uint32_t spp_begin, uint32_t count, bool sum_mode, float *output, uint32_t width, uint32_t height);
    // rays, node records, triangle records, paths, shading surfaces, traversal stack entries dropped (0 = all traversals complete)
    // One sample per pixel exactly as kernel::PathTracer::Render(const Params &) takes it (src/kernels/integrator/path.cuh:8-22):
    // device `output` of width * height float4 (running mean, row 0 = top), 1-based sample number, max_depth as passed to the
    // kernel (0xffffffff = unbounded), channel 0 colour / 1 normal.  Blocks until the sample is in the buffer.
    void RenderOneSample(float *output, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth, int channel);
    void ReadCounters(unsigned long long out[6]);
    // `count` Update() calls bracketed by events on the tracer's stream; returns device milliseconds
    float UpdateTimed(uint32_t count);
    // device time between two points of the tracer's stream: TimerBegin records, TimerEnd records, waits and returns ms
    void TimerBegin();
    float TimerEnd();
    void ResetCounters();

    struct AccelView { const cpt::AccelNode *nodes; const cpt::Bbox *bboxes; const uint64_t *keys; uint32_t num_primitives; };
    // mesh < 0: the top-level structure; otherwise the bottom-level structure of instance `mesh`
    AccelView Accel(int mesh) const;
    // per-stage milliseconds of the largest bottom-level build of the last BuildBuffers
    // (morton, sort, gather, karras, refit), valid after SetProfileBuild(true)
    void SetProfileBuild(bool on) { profile_build_ = on; }
    const float *BuildStageMs() const { return build_stage_ms_; }
    // the same mesh's triangle-box kernel before the build and triangle-record kernel after it (the node records come
    // out of the build itself): what has to run besides the build before a ray can be traced
    float LeafBoxesMs() const { return leaf_boxes_ms_; }
    float PackMs() const { return pack_ms_; }

    // Looks for edits made to the scene's materials (Material::SetChanged) and camera since the device tables were
    // written, re-uploads what moved and restarts the accumulation; Update and UpdateMany call it.  Returns whether
    // anything had changed.  With restart == false the device tables are brought up to date but film and sample counter
    // are left alone; the restart then happens at the next call that asks for it.
    bool RefreshEdits(bool restart = true);

private:
    void BuildAccel();
    void BuildInstancesAndLights();
    bool CameraFromScene(cpt::CameraDev &out);
    struct MaterialEntry { std::shared_ptr<Material> material; uint32_t revision; uint32_t tex[3]; };
    std::vector<MaterialEntry> material_entries_;        // in material-table order

    struct MeshBuffers {
        cpt::HipBuffer positions, normals, texcoords, indices;
        cpt::HipBuffer nodes, bboxes, keys;          // reference-layout build output
        cpt::HipBuffer packed_nodes, tris;           // traversal records
        cpt::HipBuffer wide_nodes;                   // two-level records, when built
        cpt::HipBuffer root;                         // the build leaves the root record here
        uint32_t num_triangles = 0;
    };

    Scene &scene_;
    Film &film_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_begin_ = nullptr, ev_end_ = nullptr;

    std::string capture_name_;
    int max_depth_ = -1;
    uint32_t curr_spp_ = 0;
    int display_channel_ = 0;
    uint32_t last_width_ = 0, last_height_ = 0;
    bool built_ = false, has_camera_ = false;
    bool profile_build_ = false;
    int integrator_ = 0;
    size_t scene_triangles_ = 0;
    uint32_t queue_parity_ = 0;
    int resident_blocks_ = 0;
    bool resident_scene_in_lds_ = false;     // resident_blocks_ was looked up for the kernel that keeps the records in LDS
    bool scene_in_lds_ = true;
    cpt::HipBuffer queue_;
    cpt::HipBuffer query_queue_;           // segment cursors of the persistent query kernel
    uint32_t query_parity_ = 0;
    int query_blocks_ = 0, query_key_ = -1;
    bool query_stream_ = false;
    cpt::HipBuffer order_;                 // one word per pixel for launches of several samples (RenderParams::order)
    bool restart_pending_ = false;         // an edit was uploaded by RefreshEdits(false); the accumulation has not been restarted yet
    uint64_t film_zero_generation_ = 0;    // Film::ZeroGeneration() when the two below were last brought up to date
    long long film_sum_count_ = 0;         // the film holds the sums of this many samples and nothing else (its alpha is that count everywhere); -1: not known
    bool film_alpha_unit_ = true;          // every alpha of the film is 0 or 1: nothing but running-mean samples and clears has written it
    uint32_t samples_per_launch_ = 32;        // measured on cornell-box-glass 1024 x 1024: 1 -> 1.30, 8 -> 0.90, 16 -> 0.87, 32 -> 0.85, 64 -> 0.84 ms per sample (later, with 0.824 at 32: 64 -> 0.817, 128 -> 0.814, 256 -> 0.813, 1024 -> 0.811)
    void LaunchSample(const cpt::RenderParams &p);
    void SyncFilmState();
    void LaunchSamples(cpt::RenderParams p, uint32_t count);     // samples p.spp .. p.spp + count - 1
    void LaunchWavefront(const cpt::RenderParams &p);
    cpt::HipBuffer wf_state_, wf_queues_;
    cpt::WavefrontBuffers wf_{};
    int wf_resident_blocks_ = 0, wf_finish_blocks_ = 0;
    bool wf_dense_ = false;          // set by BuildBuffers from the scene's triangle count: which build of the persistent kernels runs
    bool wf_wide_ = false;           // the wavefront kernels' resident counts were looked up for two-level records
    bool resident_dense_ = false;
    int resident_variant_ = 0;
    int resident_traversal_ = -1;      // bit 0: ordered traversal, bit 1: two-level records
    int wavefront_tail_ = 3;
    float build_stage_ms_[5] = {0, 0, 0, 0, 0};
    float leaf_boxes_ms_ = 0.0f, pack_ms_ = 0.0f;
    size_t largest_mesh_ = (size_t)-1;

    cpt::BuildWorkspace workspace_;
    std::vector<std::unique_ptr<MeshBuffers>> meshes_;
    std::unordered_map<const Mesh *, uint32_t> mesh_index_;
    std::vector<uint32_t> instance_mesh_;            // instance -> mesh table index
    cpt::HipBuffer leaf_scratch_;
    cpt::HipBuffer top_nodes_, top_bboxes_, top_keys_, top_packed_, top_root_, top_wide_;
    cpt::HipBuffer lds_scene_;       // node and triangle records of a small scene in one piece (SceneDev::lds_scene), or empty
    int node_records_ = 1;
    bool binary_records_ = false;              // the binary node records exist (always when node_records_ is 0)
    void EnsureBinaryRecords();
    cpt::HipBuffer mesh_table_, instance_table_, material_table_, light_table_, counters_;
    cpt::HipBuffer texel_pool_, texture_table_, srgb_table_;
    uint32_t num_lights_ = 0;
    cpt::SceneDev scene_dev_{};
};
