#include "multi_gpu.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <exception>
#include <mutex>
#include <thread>

#include "../render.h"

namespace cpt {

SampleBlock split_samples(uint32_t total, uint32_t rank, uint32_t world, uint32_t first) {
    if (world == 0 || rank >= world) throw std::invalid_argument("split_samples: rank outside the world");
    const uint32_t q = total / world, r = total % world;
    const uint32_t begin = first + rank * q + (rank < r ? rank : r);
    return SampleBlock{begin, q + (rank < r ? 1u : 0u)};
}

namespace {

// The entry points of librccl this file uses, with the signatures <rccl/rccl.h> declares.
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl &rccl() {
    static Rccl api;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        // a librccl.so.1 the process already holds (PyTorch ships one) is the one to share; otherwise ROCm's
        void *lib = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { failure = std::string("cannot open librccl: ") + dlerror(); return; }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(lib, name);
            if (!p && failure.empty()) failure = std::string("librccl lacks ") + name;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.CommCuDevice = reinterpret_cast<decltype(api.CommCuDevice)>(sym("ncclCommCuDevice"));
        api.Reduce = reinterpret_cast<decltype(api.Reduce)>(sym("ncclReduce"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!failure.empty()) throw std::runtime_error(failure);
    return api;
}

void rccl_check(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().GetErrorString(r));
}

// makes `device` current for a scope
struct DeviceScope {
    int previous = 0;
    explicit DeviceScope(int device) {
        CPT_HIP(hipGetDevice(&previous));
        if (device != previous) CPT_HIP(hipSetDevice(device));
    }
    ~DeviceScope() { (void)hipSetDevice(previous); }
};

}  // namespace

ShardComm::~ShardComm() {
    if (!comm_) return;
    try {
        DeviceScope scope(device_);
        (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    } catch (...) {
    }
}

std::vector<std::unique_ptr<ShardComm>> ShardComm::InitAll(const std::vector<int> &devices) {
    if (devices.empty()) throw std::invalid_argument("ShardComm::InitAll: no devices");
    for (size_t i = 0; i < devices.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (devices[i] == devices[j]) throw std::invalid_argument("ShardComm::InitAll: a device appears twice");
    std::vector<ncclComm_t> comms(devices.size());
    rccl_check(rccl().CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
    std::vector<std::unique_ptr<ShardComm>> out;
    for (size_t i = 0; i < devices.size(); i++) {
        std::unique_ptr<ShardComm> c(new ShardComm());
        c->comm_ = comms[i];
        c->rank_ = (int)i;
        c->world_ = (int)devices.size();
        c->device_ = devices[i];
        out.push_back(std::move(c));
    }
    return out;
}

std::array<char, kCommIdBytes> ShardComm::UniqueId() {
    static_assert(sizeof(ncclUniqueId) == kCommIdBytes, "id size");
    ncclUniqueId id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::array<char, kCommIdBytes> out;
    memcpy(out.data(), &id, kCommIdBytes);
    return out;
}

std::unique_ptr<ShardComm> ShardComm::InitRank(const std::array<char, kCommIdBytes> &id, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("ShardComm::InitRank: rank outside the world");
    ncclUniqueId nid;
    memcpy(&nid, id.data(), kCommIdBytes);
    std::unique_ptr<ShardComm> c(new ShardComm());
    CPT_HIP(hipGetDevice(&c->device_));
    ncclComm_t comm = nullptr;
    rccl_check(rccl().CommInitRank(&comm, world, nid, rank), "ncclCommInitRank");
    c->comm_ = comm;
    c->rank_ = rank;
    c->world_ = world;
    return c;
}

// what the communicator itself says about its size, this rank and its device (not what it was asked for)
ShardComm::CensusResult ShardComm::Census() const {
    CensusResult r{0, -1, -1};
    rccl_check(rccl().CommCount(static_cast<ncclComm_t>(comm_), &r.ranks), "ncclCommCount");
    rccl_check(rccl().CommUserRank(static_cast<ncclComm_t>(comm_), &r.rank), "ncclCommUserRank");
    rccl_check(rccl().CommCuDevice(static_cast<ncclComm_t>(comm_), &r.device), "ncclCommCuDevice");
    return r;
}

void ShardComm::ReduceSum(float *d_buffer, size_t count, int root, hipStream_t stream) {
    if (root < 0 || root >= world_) throw std::invalid_argument("ShardComm::ReduceSum: root outside the world");
    rccl_check(rccl().Reduce(d_buffer, d_buffer, count, ncclFloat, ncclSum, root, static_cast<ncclComm_t>(comm_), stream), "ncclReduce");
}

void ShardComm::GroupBegin() { rccl_check(rccl().GroupStart(), "ncclGroupStart"); }
void ShardComm::GroupEnd() { rccl_check(rccl().GroupEnd(), "ncclGroupEnd"); }

void FinishSumFilm(float *d_film, uint32_t width, uint32_t height, hipStream_t stream) {
    int rc = launch_film_finish(d_film, (uint64_t)width * height, stream);
    if (rc) throw std::runtime_error(std::string("film finish: ") + hipGetErrorString((hipError_t)rc));
}

void AddFilm(float *d_dst, const float *d_src, uint32_t width, uint32_t height, hipStream_t stream) {
    int rc = launch_film_add(d_dst, d_src, (uint64_t)width * height, stream);
    if (rc) throw std::runtime_error(std::string("film add: ") + hipGetErrorString((hipError_t)rc));
}

void ReduceFilmToRoot(PathTracer &tracer, Film &film, ShardComm &comm, int root, bool finish) {
    hipStream_t stream = tracer.Stream();
    float *buffer = film.DeviceMap(stream);
    comm.ReduceSum(buffer, (size_t)film.Width() * film.Height() * 4, root, stream);
    if (finish && comm.Rank() == root) FinishSumFilm(buffer, film.Width(), film.Height(), stream);
    tracer.NoteFilmExposed();        // the root's film now holds other ranks' samples (and means, once finished)
}

}  // namespace cpt

using cpt::DeviceScope;

MultiPathTracer::MultiPathTracer(Scene &scene, Film &film, std::vector<int> shard_devices) : scene_(scene), film_(film) {
    if (shard_devices.empty()) throw std::invalid_argument("MultiPathTracer: no shards");
    int available = 0;
    CPT_HIP(hipGetDeviceCount(&available));
    std::vector<int> distinct;
    for (size_t i = 0; i < shard_devices.size(); i++) {
        const int d = shard_devices[i];
        if (d < 0 || d >= available)
            throw std::invalid_argument("MultiPathTracer: device " + std::to_string(d) + " of " + std::to_string(available) + " visible");
        Shard s;
        s.device = d;
        s.leader = (int)i;
        for (size_t j = 0; j < i; j++)
            if (shards_[j].device == d) { s.leader = shards_[j].leader; break; }
        if (s.leader == (int)i) { distinct.push_back(d); comm_shard_.push_back((int)i); }
        shards_.push_back(std::move(s));
    }
    comms_ = cpt::ShardComm::InitAll(distinct);
    for (size_t r = 0; r < comms_.size(); r++) {
        const cpt::ShardComm::CensusResult seen = comms_[r]->Census();
        if (seen.ranks != (int)distinct.size() || seen.rank != (int)r || seen.device != distinct[r])
            throw std::runtime_error("MultiPathTracer: RCCL reports " + std::to_string(seen.ranks) + " ranks, rank " + std::to_string(seen.rank) + " on device " +
                                     std::to_string(seen.device) + " where rank " + std::to_string(r) + " of " + std::to_string(distinct.size()) + " on device " +
                                     std::to_string(distinct[r]) + " was asked for");
    }
    for (size_t i = 0; i < shards_.size(); i++) {
        Shard &s = shards_[i];
        DeviceScope scope(s.device);
        if (i == 0) {
            s.film = &film_;
        } else {
            s.own_film = std::make_unique<Film>(film_.Width(), film_.Height());
            s.film = s.own_film.get();
        }
        // a Film takes its device memory at first use, on whichever device is current then: take it here, on the
        // shard's own device, the caller's film of shard 0 included
        (void)s.film->DeviceMap(nullptr);
        s.tracer = std::make_unique<PathTracer>(scene_, *s.film);
        CPT_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
}

MultiPathTracer::~MultiPathTracer() {
    for (Shard &s : shards_) {
        try {
            DeviceScope scope(s.device);
            if (s.tracer) s.tracer->Synchronize();
            s.tracer.reset();
            s.own_film.reset();
            if (s.done) (void)hipEventDestroy(s.done);
        } catch (...) {
        }
    }
}

// fn(shard index) on one host thread per shard, that shard's device current; the first exception is rethrown here
template <typename F> void MultiPathTracer::OnEveryShard(F &&fn) {
    std::vector<std::thread> threads;
    std::exception_ptr failure;
    std::mutex failure_mutex;
    for (size_t i = 0; i < shards_.size(); i++) {
        threads.emplace_back([&, i] {
            try {
                CPT_HIP(hipSetDevice(shards_[i].device));
                fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lock(failure_mutex);
                if (!failure) failure = std::current_exception();
            }
        });
    }
    for (auto &t : threads) t.join();
    if (failure) std::rethrow_exception(failure);
}

void MultiPathTracer::BuildBuffers() {
    OnEveryShard([&](size_t i) {
        PathTracer &t = *shards_[i].tracer;
        t.SetMaxDepth(max_depth_);
        t.SetIntegrator(integrator_);
        t.SetChannel(channel_);
        t.SetTraversal(traversal_);
        t.SetSamplesPerLaunch(samples_per_launch_);
        t.SetWavefrontTail(wavefront_tail_);
        t.SetNodeRecords(node_records_);
        t.BuildBuffers();
    });
    built_ = true;
}

uint32_t MultiPathTracer::NumLights() const { return shards_.front().tracer->NumLights(); }

void MultiPathTracer::Render(uint32_t total_spp, uint32_t first, void (*progress)(uint32_t, void *), void *user) {
    if (!built_) throw std::runtime_error("MultiPathTracer::Render before BuildBuffers");
    for (const Shard &s : shards_)
        if (s.film->Width() != film_.Width() || s.film->Height() != film_.Height()) throw std::runtime_error("MultiPathTracer: film resized");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t world = (uint32_t)shards_.size();
    std::mutex progress_mutex;
    uint32_t done_total = 0;
    // every shard: clear its film, queue its block of samples as sums, mark the end on its stream
    OnEveryShard([&](size_t i) {
        Shard &s = shards_[i];
        PathTracer &t = *s.tracer;
        t.SetMaxDepth(max_depth_);
        t.SetIntegrator(integrator_);
        t.SetChannel(channel_);
        t.SetTraversal(traversal_);
        t.SetSamplesPerLaunch(samples_per_launch_);
        t.SetWavefrontTail(wavefront_tail_);
        s.film->Clear(t.Stream());
        t.NoteFilmCleared();
        const cpt::SampleBlock block = cpt::split_samples(total_spp, (uint32_t)i, world, first);
        const uint32_t step = progress ? 16u : (block.count ? block.count : 1u);
        for (uint32_t at = 0; at < block.count; at += step) {
            const uint32_t n = std::min(step, block.count - at);
            t.RenderSamples(block.begin + at, n, true, nullptr, 0, 0);
            if (progress) {
                t.Synchronize();
                std::lock_guard<std::mutex> lock(progress_mutex);
                done_total += n;
                progress(done_total, user);
            }
        }
        CPT_HIP(hipEventRecord(s.done, t.Stream()));
    });
    // shards that share a device: summed into the device's first shard, on that shard's stream
    for (size_t i = 0; i < shards_.size(); i++) {
        Shard &s = shards_[i];
        if (s.leader == (int)i) continue;
        Shard &lead = shards_[s.leader];
        DeviceScope scope(s.device);
        hipStream_t stream = lead.tracer->Stream();
        CPT_HIP(hipStreamWaitEvent(stream, s.done, 0));
        cpt::AddFilm(lead.film->DeviceMap(stream), s.film->DeviceMap(s.tracer->Stream()), film_.Width(), film_.Height(), stream);
    }
    // one sum-reduce between the devices, each rank on its leader's stream (ordered after that device's renders)
    const size_t count = (size_t)film_.Width() * film_.Height() * 4;
    cpt::ShardComm::GroupBegin();
    for (size_t r = 0; r < comms_.size(); r++) {
        Shard &lead = shards_[comm_shard_[r]];
        DeviceScope scope(lead.device);
        comms_[r]->ReduceSum(lead.film->DeviceMap(lead.tracer->Stream()), count, 0, lead.tracer->Stream());
    }
    cpt::ShardComm::GroupEnd();
    {
        Shard &root = shards_.front();
        DeviceScope scope(root.device);
        cpt::FinishSumFilm(film_.DeviceMap(root.tracer->Stream()), film_.Width(), film_.Height(), root.tracer->Stream());
    }
    for (Shard &s : shards_) s.tracer->NoteFilmExposed();       // the films hold other shards' sums now
    Synchronize();
    last_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void MultiPathTracer::Synchronize() {
    for (Shard &s : shards_) {
        DeviceScope scope(s.device);
        s.tracer->Synchronize();
    }
}

void MultiPathTracer::ReadCounters(unsigned long long out[6]) {
    for (int k = 0; k < 6; k++) out[k] = 0;
    for (Shard &s : shards_) {
        DeviceScope scope(s.device);
        unsigned long long c[6];
        s.tracer->ReadCounters(c);
        for (int k = 0; k < 6; k++) out[k] += c[k];
    }
}

void MultiPathTracer::ResetCounters() {
    for (Shard &s : shards_) {
        DeviceScope scope(s.device);
        s.tracer->ResetCounters();
    }
}
