// Command-line renderer.  Same positional arguments, flags, defaults and exit codes as the reference's
// executable on its headless path (src/main.cpp:18-48, 121-130):
//   cuda-pathtracer <obj> <json|xml> [--ui 0|1] [--max-depth N] [--max-spp N] [--output|-o NAME]
//                   [--bsdf-type lambert|phong|blinn-phong|microfacet]
// plus four flags of this build: --gpus N (render on N GPUs of this node: replicated scene, samples split,
// one RCCL sum-reduce), --shards N (logical shards, default = gpus; shard i runs on GPU i mod gpus),
// --scene-cache 0|1 (keep a binary copy of the parsed scene next to the .obj) and --timings 0|1 (print the seconds the
// scene load, the device build, the first sample and the whole run took).  There is no interactive
// window: --ui is accepted, 1 only prints a note.  Writes NAME.exr.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <functional>
#include <iostream>
#include <string>
#include <string_view>
#include <vector>

#include "host/film.hpp"
#include "host/loader.hpp"
#include "host/multi_gpu.hpp"
#include "host/pathtracer.hpp"

namespace {

struct Options {
    int ui = 1;
    int max_depth = -1;
    int max_spp = -1;
    std::string output = "capture";
    std::string bsdf = "blinn-phong";
    int gpus = 1;
    int shards = 0;            // 0 = one per GPU
    int scene_cache = 0;
    int timings = 0;
};

struct Flag {
    std::vector<std::string_view> spellings;
    const char *help;
    std::function<void(Options &, const char *)> store;
};

const std::vector<Flag> &Flags() {
    static const std::vector<Flag> table = {
        {{"--ui"}, "0 or 1; this build has no window and always renders headless", [](Options &o, const char *v) { o.ui = std::atoi(v); }},
        {{"--max-depth"}, "longest path in bounces; -1 (default) leaves termination to Russian roulette", [](Options &o, const char *v) { o.max_depth = std::atoi(v); }},
        {{"--max-spp"}, "samples per pixel to render; -1 (default) means 65536", [](Options &o, const char *v) { o.max_spp = std::atoi(v); }},
        {{"--output", "-o"}, "name of the image, written as NAME.exr (default: capture)", [](Options &o, const char *v) { o.output = v; }},
        {{"--bsdf-type"}, "lambert, phong, blinn-phong (default) or microfacet", [](Options &o, const char *v) { o.bsdf = v; }},
        {{"--gpus"}, "GPUs of this node to render on (default 1): scene replicated, samples split, one RCCL reduce", [](Options &o, const char *v) { o.gpus = std::atoi(v); }},
        {{"--shards"}, "sample blocks rendered side by side (default: one per GPU); shard i runs on GPU i mod gpus", [](Options &o, const char *v) { o.shards = std::atoi(v); }},
        {{"--scene-cache"}, "1: read and keep <obj>.cptcache, a binary copy of the parsed scene (default 0)", [](Options &o, const char *v) { o.scene_cache = std::atoi(v); }},
        {{"--timings"}, "1: print the seconds spent loading the scene, building on the device, on the first sample and in all (default 0)", [](Options &o, const char *v) { o.timings = std::atoi(v); }},
    };
    return table;
}

int Usage(const char *program) {
    std::cout << "usage: " << program << " <scene.obj> <camera.json | scene.xml> [flags]\n\nflags (each takes one value):\n";
    for (const Flag &f : Flags()) {
        std::string names;
        for (std::string_view s : f.spellings) names += (names.empty() ? "" : ", ") + std::string(s);
        std::cout << "  " << names << std::string(names.size() < 16 ? 16 - names.size() : 1, ' ') << f.help << "\n";
    }
    return -1;
}

// unknown words and a flag without its value are skipped, as the reference skips what it does not know
void ParseFlags(int argc, char **argv, Options &options) {
    for (int i = 3; i < argc; i++) {
        for (const Flag &f : Flags()) {
            if (std::find(f.spellings.begin(), f.spellings.end(), std::string_view(argv[i])) == f.spellings.end()) continue;
            if (i + 1 < argc) f.store(options, argv[++i]);
            break;
        }
    }
}

int BsdfNumber(const std::string &name) {
    static const char *const known[] = {"lambert", "phong", "blinn-phong", "microfacet"};
    for (int k = 0; k < 4; k++)
        if (name == known[k]) return k;
    return 2;                  // an unknown name keeps the default, blinn-phong (src/main.cpp:41-48)
}

int Complain(const std::string &message) {
    std::cout << message << std::endl;
    return -1;
}

// called (serialised) with the samples finished so far over all shards: one line per multiple of 16 passed
void ReportProgress(uint32_t done, void *state) {
    uint32_t &reported = *static_cast<uint32_t *>(state);
    while (reported + 16 <= done) {
        reported += 16;
        std::cout << "spp " << reported << std::endl;
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) return Usage(argv[0]);
    Options options;
    ParseFlags(argc, argv, options);

    const std::filesystem::path obj_path(argv[1]), extra_path(argv[2]);
    if (!std::filesystem::exists(obj_path)) return Complain("obj file '" + obj_path.string() + "' doesn't exist");
    if (!std::filesystem::exists(extra_path)) return Complain("extra file '" + extra_path.string() + "' doesn't exist");
    const std::string extension = extra_path.extension().string();
    if (extension != ".json" && extension != ".xml") return Complain("unknown extra file extension '" + extension + "'");
    if (options.gpus < 1 || options.shards < 0) return Complain("--gpus needs a positive number, --shards a non-negative one");

    const auto started = std::chrono::steady_clock::now();
    auto seconds_since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count(); };
    try {
        Scene scene{};
        bool from_cache = false;
        const int bsdf = BsdfNumber(options.bsdf);
        const bool loaded = options.scene_cache ? LoadObjSceneCached(scene, obj_path, bsdf, &from_cache) : LoadObjScene(scene, obj_path, bsdf);
        if (!loaded) return Complain("failed to load obj file '" + obj_path.string() + "'");
        if (from_cache) std::cout << "scene read from '" << obj_path.string() << ".cptcache'" << std::endl;
        const bool extra_ok = extension == ".json" ? LoadJsonScene(scene, extra_path) : LoadXmlScene(scene, extra_path);
        if (!extra_ok) return Complain("failed to load " + extension.substr(1) + " file '" + extra_path.string() + "'");
        if (options.ui) std::cout << "note: no interactive window in this build, rendering headless" << std::endl;
        if (options.timings) std::cout << "timing: scene loaded after " << seconds_since_start() << " s" << (from_cache ? " (from the cache)" : "") << std::endl;

        auto camera_object = scene.FirstObjectWith<CameraComponent>();
        if (!camera_object) return Complain("the scene has no camera");
        const CameraComponent *camera = camera_object->GetComponent<CameraComponent>();
        Film film(camera->film_width, camera->film_height);
        const uint32_t max_spp = std::min(static_cast<uint32_t>(options.max_spp), 65536u);       // -1 wraps to the cap (src/main.cpp:122)
        const std::string exr_name = options.output + ".exr";

        const int shards = options.shards ? options.shards : options.gpus;
        if (options.gpus == 1 && shards == 1) {
            auto path_tracer = camera_object->AddComponent<PathTracer>(scene, film);
            path_tracer->SetMaxDepth(options.max_depth);
            path_tracer->SetIntegrator(3);                // bounce-loop kernel by scene size; the image does not depend on it
            path_tracer->SetCaptureName(options.output);
            path_tracer->BuildBuffers();
            if (options.timings) std::cout << "timing: device buffers and BVHs built after " << seconds_since_start() << " s" << std::endl;
            if (options.timings && max_spp != 0) {       // the first sample on its own, so that its end can be reported
                path_tracer->UpdateMany(1);
                path_tracer->Synchronize();
                std::cout << "timing: first sample finished after " << seconds_since_start() << " s" << std::endl;
            }
            // The reference renders one sample per scene.Update() and reports every 16th (src/main.cpp:123-128);
            // the samples between two reports are asked for together, which leaves the same film.
            for (uint32_t done = (options.timings && max_spp != 0) ? 1 : 0; done < max_spp;) {
                const uint32_t batch = std::min(16u - done % 16u, max_spp - done);
                path_tracer->UpdateMany(batch);
                done += batch;
                if (done % 16 == 0) {
                    path_tracer->Synchronize();
                    std::cout << "spp " << done << std::endl;
                }
            }
            path_tracer->Synchronize();
        } else {
            int visible = 0;
            CPT_HIP(hipGetDeviceCount(&visible));
            if (options.gpus > visible) return Complain("--gpus " + std::to_string(options.gpus) + ", but " + std::to_string(visible) + " visible");
            std::vector<int> devices;
            for (int s = 0; s < shards; s++) devices.push_back(s % options.gpus);
            MultiPathTracer tracer(scene, film, devices);
            tracer.SetMaxDepth(options.max_depth);
            tracer.SetIntegrator(3);
            tracer.BuildBuffers();
            uint32_t reported = 0;
            if (tracer.NumLights() != 0) tracer.Render(max_spp, 1, ReportProgress, &reported);
            std::cout << shards << " shards on " << tracer.NumDevices() << " GPU(s): " << tracer.LastRenderMs() << " ms (RCCL communicator of " << tracer.RcclRanks() << " rank(s))" << std::endl;
        }
        if (options.timings) std::cout << "timing: rendered after " << seconds_since_start() << " s" << std::endl;
        if (!film.SaveTo(exr_name.c_str())) return Complain("failed to write '" + exr_name + "'");
        if (options.timings) std::cout << "timing: image written after " << seconds_since_start() << " s" << std::endl;
    } catch (const std::exception &e) {
        return Complain(std::string("error: ") + e.what());
    }
    return 0;
}
