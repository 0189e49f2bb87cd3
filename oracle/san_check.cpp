// TEST INFRASTRUCTURE: host-side code (the host ingest library and the CPU oracle) compiled with AddressSanitizer and
// UndefinedBehaviorSanitizer into one executable and run over a fixture: read the OBJ, build the tree, render two small
// frames (both invDir modes, a debug mode), query primary hits with and without the BVH.  Any report fails the run.
// usage: san_check <scene.obj>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_api.h"
#include "rt_host.h"

extern "C" {
void oracle_render(const rt_scene* s, const rt_camera* cam, const rt_params* p, float* rgba, unsigned long long* counters,
                   int num_threads);
void oracle_primary_hits(const rt_scene* s, const rt_camera* cam, int width, int height, int brute_force, int32_t* ids, float* t,
                         int num_threads);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    rt_host_scene hs;
    if (rt_host_load_scene(argv[1], &hs) != 0) {
        std::fprintf(stderr, "cannot load %s\n", argv[1]);
        return 1;
    }
    std::vector<float> colours((size_t)hs.num_tris * 4);
    rt_host_make_colours(1u, hs.num_tris, colours.data());
    rt_scene sc{hs.num_nodes, hs.num_tris, hs.nodes, hs.triv0, hs.triv1, hs.triv2, colours.data(), hs.links};
    rt_camera cam{};
    cam.pos[2] = -2.0f;
    cam.rot[0] = cam.rot[4] = cam.rot[8] = 1.0f;
    cam.fov_rad = 1.57079637f;
    cam.viewport_dist = 0.1f;
    const int W = 64, H = 36;
    std::vector<float> img((size_t)W * H * 4);
    unsigned long long counters[RT_NUM_COUNTERS] = {0};
    double sum = 0.0;
    const uint32_t variants[3][2] = {{RT_MODE_RENDER, 0u}, {RT_MODE_RENDER, RT_FLAG_FIX_INVDIR}, {RT_MODE_BOX_TESTS, 0u}};
    for (const auto& v : variants) {
        rt_params p{W, H, 1u, 100u, v[0], 0u, 2u, 1u, v[1], 0u};
        oracle_render(&sc, &cam, &p, img.data(), counters, 2);
        for (float x : img) sum += x;
    }
    std::vector<int32_t> ids((size_t)W * H), ids_bf((size_t)W * H);
    std::vector<float> t((size_t)W * H), t_bf((size_t)W * H);
    oracle_primary_hits(&sc, &cam, W, H, 0, ids.data(), t.data(), 2);
    oracle_primary_hits(&sc, &cam, W, H, 1, ids_bf.data(), t_bf.data(), 2);
    int hits = 0;
    for (int32_t i : ids) hits += i >= 0;
    std::printf("san_check ok: %d triangles, %d nodes, %d primary hits, %llu rays, checksum %.3f\n", hs.num_tris, hs.num_nodes, hits,
                counters[2], sum);
    rt_host_free_scene(&hs);
    return 0;
}
