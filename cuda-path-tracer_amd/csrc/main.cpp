// Command-line renderer with the reference's interface (src/main.cpp:8-134), headless only:
//   cuda-pathtracer <obj> <json|xml> [--ui 0|1] [--max-depth N] [--max-spp N] [--output|-o NAME]
//                   [--bsdf-type lambert|phong|blinn-phong|microfacet] [--scene-cache 0|1]
// The interactive window (--ui 1) does not exist in this build; the flag is accepted and ignored
// with a note, everything else behaves as the reference's --ui 0 path and writes NAME.exr.
// --scene-cache 1 (not in the reference) keeps a binary copy of the parsed scene next to the .obj.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>

#include "host/film.hpp"
#include "host/loader.hpp"
#include "host/pathtracer.hpp"

int main(int argc, char **argv) {
    struct {
        bool ui = true;
        int max_depth = -1;
        int max_spp = -1;
        const char *capture_name = "capture";
        const char *bsdf_type = "blinn-phong";
        int bsdf_type_i = 2;
        bool scene_cache = false;
    } cmd_args;

    if (argc < 3) {
        std::cout << "usage: " << argv[0] << " <path-to-obj-file> <path-to-extra-file> [OPTIONS]\n";
        std::cout << "\nOPTIONS\n";
        std::cout << "  --ui            0 or 1, accepted for compatibility; this build always runs headless\n";
        std::cout << "  --max-depth     max ray tracing depth (default -1, no limit)\n";
        std::cout << "  --max-spp       max ray tracing spp (default -1, which renders 65536)\n";
        std::cout << "  --output | -o   output .exr name (default 'capture')\n";
        std::cout << "  --bsdf-type     which BSDF to use (default 'blinn-phong')\n";
        std::cout << "  --scene-cache   0 or 1 (default 0): keep and reuse <obj>.cptcache, a binary copy of the parsed scene\n";
        return -1;
    }
    for (int i = 3; i < argc; i++) {
        bool has_value = i + 1 < argc;
        if (strcmp(argv[i], "--ui") == 0 && has_value) {
            cmd_args.ui = std::atoi(argv[++i]);
        } else if (strcmp(argv[i], "--max-depth") == 0 && has_value) {
            cmd_args.max_depth = std::atoi(argv[++i]);
        } else if (strcmp(argv[i], "--max-spp") == 0 && has_value) {
            cmd_args.max_spp = std::atoi(argv[++i]);
        } else if ((strcmp(argv[i], "--output") == 0 || strcmp(argv[i], "-o") == 0) && has_value) {
            cmd_args.capture_name = argv[++i];
        } else if (strcmp(argv[i], "--bsdf-type") == 0 && has_value) {
            cmd_args.bsdf_type = argv[++i];
        } else if (strcmp(argv[i], "--scene-cache") == 0 && has_value) {
            cmd_args.scene_cache = std::atoi(argv[++i]);
        }
    }
    const char *bsdf_type_names[] = {"lambert", "phong", "blinn-phong", "microfacet"};
    for (int i = 0; i < 4; i++)
        if (strcmp(cmd_args.bsdf_type, bsdf_type_names[i]) == 0) cmd_args.bsdf_type_i = i;

    std::filesystem::path obj_path(argv[1]);
    if (!std::filesystem::exists(obj_path)) {
        std::cout << "obj file '" << obj_path << "' doesn't exist" << std::endl;
        return -1;
    }
    std::filesystem::path extra_path(argv[2]);
    if (!std::filesystem::exists(extra_path)) {
        std::cout << "extra file '" << extra_path << "' doesn't exist" << std::endl;
        return -1;
    }

    Scene scene{};
    bool from_cache = false;
    if (!(cmd_args.scene_cache ? LoadObjSceneCached(scene, obj_path, cmd_args.bsdf_type_i, &from_cache) : LoadObjScene(scene, obj_path, cmd_args.bsdf_type_i))) {
        std::cout << "failed to load obj file '" << obj_path << "'" << std::endl;
        return -1;
    }
    if (from_cache) std::cout << "scene read from '" << obj_path.string() << ".cptcache'" << std::endl;
    if (extra_path.extension() == ".json") {
        if (!LoadJsonScene(scene, extra_path)) {
            std::cout << "failed to load json file '" << extra_path << "'" << std::endl;
            return -1;
        }
    } else if (extra_path.extension() == ".xml") {
        if (!LoadXmlScene(scene, extra_path)) {
            std::cout << "failed to load xml file '" << extra_path << "'" << std::endl;
            return -1;
        }
    } else {
        std::cout << "unknown extra file extension '" << extra_path.extension().string() << "'" << std::endl;
        return -1;
    }
    if (cmd_args.ui) std::cout << "note: no interactive window in this build, rendering headless" << std::endl;

    try {
        auto camera_object = scene.FirstObjectWith<CameraComponent>();
        auto camera_comp = camera_object->GetComponent<CameraComponent>();
        Film film(camera_comp->film_width, camera_comp->film_height);

        auto path_tracer = camera_object->AddComponent<PathTracer>(scene, film);
        path_tracer->SetMaxDepth(cmd_args.max_depth);
        path_tracer->SetIntegrator(3);       // bounce-loop variant by scene size; the image does not depend on it
        path_tracer->SetCaptureName(cmd_args.capture_name);
        path_tracer->BuildBuffers();

        uint32_t max_spp = std::min(static_cast<uint32_t>(cmd_args.max_spp), 65536u);
        // the reference calls scene.Update() once per sample and reports every 16th; the samples between two
        // reports are asked for together here, which leaves the same film (PathTracer::UpdateMany)
        for (uint32_t done = 0; done < max_spp;) {
            const uint32_t batch = std::min(16u - done % 16u, max_spp - done);
            path_tracer->UpdateMany(batch);
            done += batch;
            if (done % 16 == 0) {
                path_tracer->Synchronize();
                std::cout << "spp " << done << std::endl;
            }
        }
        path_tracer->Synchronize();
        auto exr_name = std::string(cmd_args.capture_name) + ".exr";
        if (!film.SaveTo(exr_name.c_str())) {
            std::cout << "failed to write '" << exr_name << "'" << std::endl;
            return -1;
        }
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << std::endl;
        return -1;
    }
    return 0;
}
