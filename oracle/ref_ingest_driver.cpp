// TEST INFRASTRUCTURE.  Driver around the REFERENCE's own host objects (obj-reader.cpp, bvh.cpp,
// scene-loader.cpp), which oracle/Makefile compiles from /root/reference where they lie; nothing of
// the reference is copied into this repository.  It runs the reference's loadScene() on a given OBJ
// and dumps the resulting buffers, plus the outputs of the two other host-side formulas the render
// path's inputs depend on (libstdc++'s uniform_real_distribution colours of raytrace.cpp:44-52 and the
// GLM camera matrix of main.cpp:41-51), so that tests can compare librt_host.so byte for byte.
//
// usage: ref_ingest <scene.obj> <out.bin> [colour_seed] [pitch yaw]
//
// loadScene() reads the compiled-in relative path "../models/teapot.obj" (constants.h:14), so the
// driver stages the OBJ under that name next to a scratch working directory inside oracle/_ref/.

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <random>
#include <string>
#include <vector>

#include <unistd.h>

#include "glm/glm.hpp"
#include "scene-loader.h"

namespace fs = std::filesystem;

static void put(FILE* f, const void* p, size_t n) {
    if (n && std::fwrite(p, 1, n, f) != n) {
        std::perror("write");
        std::exit(2);
    }
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <scene.obj> <out.bin> [colour_seed] [pitch yaw]\n", argv[0]);
        return 1;
    }
    fs::path obj = fs::absolute(argv[1]);
    fs::path out = fs::absolute(argv[2]);
    uint32_t seed = argc > 3 ? (uint32_t)std::strtoul(argv[3], nullptr, 10) : 1u;
    double pitch = argc > 5 ? std::atof(argv[4]) : 0.0;
    double yaw = argc > 5 ? std::atof(argv[5]) : 0.0;

    fs::path stage = out.parent_path() / ("stage_" + std::to_string((long)getpid()));
    fs::create_directories(stage / "models");
    fs::create_directories(stage / "cwd");
    fs::copy_file(obj, stage / "models" / "teapot.obj", fs::copy_options::overwrite_existing);
    fs::current_path(stage / "cwd");

    SceneData* scene = loadScene();

    fs::current_path(out.parent_path());
    fs::remove_all(stage);

    FILE* f = std::fopen(out.string().c_str(), "wb");
    if (!f) {
        std::perror("open output");
        return 2;
    }
    int32_t hdr[4] = {scene->numTriangles, (int32_t)scene->alignedBVHNodes.size(), (int32_t)sizeof(AlignedMat3), 0};
    put(f, hdr, sizeof hdr);
    put(f, scene->alignedBVHNodes.data(), scene->alignedBVHNodes.size() * sizeof(AlignedMat3));
    put(f, scene->triv0.data(), scene->triv0.size() * sizeof(glm::vec4));
    put(f, scene->triv1.data(), scene->triv1.size() * sizeof(glm::vec4));
    put(f, scene->triv2.data(), scene->triv2.size() * sizeof(glm::vec4));

    // colours exactly as raytrace.cpp:44-51 draws them, with a fixed seed instead of random_device
    {
        std::vector<glm::vec4> colours((size_t)scene->numTriangles);
        std::mt19937 gen(seed);
        std::uniform_real_distribution<float> dist(0.0f, 1.0f);
        for (int i = 0; i < scene->numTriangles; i++) {
            colours[i] = glm::vec4(dist(gen), dist(gen), dist(gen), 1.0f);
        }
        put(f, colours.data(), colours.size() * sizeof(glm::vec4));
    }
    // camera matrix with GLM's own mat3 product, columns as main.cpp:41-51 passes them
    {
        const float cp = (float)std::cos(pitch), sp = (float)std::sin(pitch);
        const float cy = (float)std::cos(yaw), sy = (float)std::sin(yaw);
        glm::mat3 aboutX(glm::vec3(1.0f, 0.0f, 0.0f), glm::vec3(0.0f, cp, -sp), glm::vec3(0.0f, sp, cp));
        glm::mat3 aboutY(glm::vec3(cy, 0.0f, sy), glm::vec3(0.0f, 1.0f, 0.0f), glm::vec3(-sy, 0.0f, cy));
        glm::mat3 rot = aboutY * aboutX;
        put(f, &rot[0][0], 9 * sizeof(float));
    }
    std::fclose(f);
    return 0;
}
