// Rendering on several GPUs of one node.  The reference is a single-GPU program (nothing under
// src/ names a second device); this is the partition its sampler allows: every (pixel, sample index)
// pair owns an independent random stream (src/kernels/integrator/path.cu:92 seeds the sampler from
// the pixel index and the 1-based sample number), so the samples of a render split into disjoint
// blocks, one per shard, with the scene and its BVH replicated on every GPU.  Each shard accumulates
// plain sums into its own film; ONE sum-reduce of the W*H*4 floats to the root over RCCL (xGMI
// between the GPUs) and a division by the sample count end the job.  No other exchange.
//
// Two ways to run it:
//   * one process, one host thread per shard: MultiPathTracer (what `cuda-pathtracer --gpus N` uses);
//   * one process per GPU (a launcher such as torchrun): every process owns a PathTracer and a
//     ShardComm made from a shared 128-byte id, renders its block with PathTracer::RenderSamples
//     in sum mode and calls ReduceFilmToRoot.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "film.hpp"
#include "pathtracer.hpp"
#include "scene.hpp"

namespace cpt {

// Block of 1-based sample indices shard `rank` of `world` renders when `total` samples starting at `first` are
// split: contiguous, in rank order, the remainder going to the lowest ranks (count may be 0).
struct SampleBlock { uint32_t begin, count; };
SampleBlock split_samples(uint32_t total, uint32_t rank, uint32_t world, uint32_t first = 1);

constexpr size_t kCommIdBytes = 128;     // NCCL_UNIQUE_ID_BYTES

// One rank of an RCCL communicator, bound to the HIP device that was current when it was made.
// librccl is opened when the first communicator is asked for, not when this library is loaded.
class ShardComm {
public:
    ~ShardComm();
    ShardComm(const ShardComm &) = delete;
    ShardComm &operator=(const ShardComm &) = delete;

    // single process: one rank per entry of `devices` (all different), rank i on devices[i]
    static std::vector<std::unique_ptr<ShardComm>> InitAll(const std::vector<int> &devices);
    // several processes: rank 0 makes the id, everyone gets a copy by whatever means the launcher has
    static std::array<char, kCommIdBytes> UniqueId();
    static std::unique_ptr<ShardComm> InitRank(const std::array<char, kCommIdBytes> &id, int rank, int world);

    int Rank() const { return rank_; }
    int World() const { return world_; }
    int Device() const { return device_; }
    // ncclCommCount / ncclCommUserRank / ncclCommCuDevice of this communicator: what RCCL made, not what was asked for
    struct CensusResult { int ranks, rank, device; };
    CensusResult Census() const;
    // in-place sum of `count` floats over all ranks into `root`'s buffer, ordered on `stream`
    void ReduceSum(float *d_buffer, size_t count, int root, hipStream_t stream);
    // several ranks of one process must be called between these two
    static void GroupBegin();
    static void GroupEnd();

private:
    ShardComm() = default;
    void *comm_ = nullptr;
    int rank_ = 0, world_ = 1, device_ = 0;
};

// film (sums, alpha = number of samples) -> mean radiance with alpha 1, in place, on `stream`
void FinishSumFilm(float *d_film, uint32_t width, uint32_t height, hipStream_t stream);
// d_dst += d_src for W*H*4 floats, on `stream`
void AddFilm(float *d_dst, const float *d_src, uint32_t width, uint32_t height, hipStream_t stream);

// The per-process half of a sharded render: sums `film` (sum mode) of every rank into `root`'s and, when
// `finish`, turns the root's sums into the mean.  Ordered on the tracer's stream; does not synchronise.
void ReduceFilmToRoot(PathTracer &tracer, Film &film, ShardComm &comm, int root, bool finish);

}  // namespace cpt

// One render spread over several shards, each a full PathTracer (replicated build) on its own host thread.
// shard_devices[i] is the HIP device of shard i; a device may appear more than once (the shards on it are summed
// on that device before the RCCL reduce between devices).  `film` is the result, on shard_devices[0].
class MultiPathTracer {
public:
    MultiPathTracer(Scene &scene, Film &film, std::vector<int> shard_devices);
    ~MultiPathTracer();

    void SetMaxDepth(int max_depth) { max_depth_ = max_depth; }
    void SetIntegrator(int mode) { integrator_ = mode; }
    void SetChannel(int channel) { channel_ = channel; }
    void SetTraversal(int mode) { traversal_ = mode; }
    void SetNodeRecords(int mode) { node_records_ = mode; }
    void SetSamplesPerLaunch(uint32_t samples) { samples_per_launch_ = samples; }
    void SetWavefrontTail(int bounces) { wavefront_tail_ = bounces; }

    void BuildBuffers();                    // every shard builds its replica, in parallel
    // Samples first .. first + total - 1 of every pixel, split over the shards; afterwards the film holds their
    // mean (alpha 1).  Calls `progress(done)` (may be null) from the shard threads as sample batches are queued.
    void Render(uint32_t total_spp, uint32_t first = 1, void (*progress)(uint32_t done, void *user) = nullptr, void *user = nullptr);
    void Synchronize();
    float LastRenderMs() const { return last_ms_; }          // wall time of the last Render, reduce included
    size_t NumShards() const { return shards_.size(); }
    size_t NumDevices() const { return comms_.size(); }
    int RcclRanks() const { return comms_.empty() ? 0 : comms_.front()->Census().ranks; }      // what ncclCommCount reports
    uint32_t NumLights() const;
    void ReadCounters(unsigned long long out[6]);            // summed over shards
    void ResetCounters();

private:
    struct Shard {
        int device = 0;
        int leader = 0;                     // first shard on the same device
        std::unique_ptr<Film> own_film;     // null for shard 0, which renders into the caller's film
        Film *film = nullptr;
        std::unique_ptr<PathTracer> tracer;
        hipEvent_t done = nullptr;
    };
    template <typename F> void OnEveryShard(F &&fn);

    Scene &scene_;
    Film &film_;
    std::vector<Shard> shards_;
    std::vector<std::unique_ptr<cpt::ShardComm>> comms_;     // one per distinct device, in order of first appearance
    std::vector<int> comm_shard_;                            // comm rank -> leader shard
    int max_depth_ = -1, integrator_ = 3, channel_ = 0, wavefront_tail_ = 3, traversal_ = 0, node_records_ = 1;
    uint32_t samples_per_launch_ = 32;
    bool built_ = false;
    float last_ms_ = 0.0f;
};
