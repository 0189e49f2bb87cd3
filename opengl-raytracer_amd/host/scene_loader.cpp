#include "scene_loader.h"

#include <cstring>
#include <stdexcept>

#include "rt_host.h"

SceneData* loadScene(const std::string& objPath, unsigned colourSeed) {
    rt_host_scene hs;
    int rc = rt_host_load_scene(objPath.c_str(), &hs);
    if (rc == RT_HOST_EIO) throw std::runtime_error("Could not open file: " + objPath);
    if (rc != RT_HOST_OK) throw std::runtime_error("Could not build scene from " + objPath + " (code " + std::to_string(rc) + ")");
    auto* s = new SceneData();
    s->numTriangles = hs.num_tris;
    s->alignedBVHNodes.resize((size_t)hs.num_nodes);
    std::memcpy(s->alignedBVHNodes.data(), hs.nodes, sizeof(float) * 12 * (size_t)hs.num_nodes);
    for (auto* dst : {&s->triv0, &s->triv1, &s->triv2}) dst->resize((size_t)hs.num_tris);
    std::memcpy(s->triv0.data(), hs.triv0, sizeof(float) * 4 * (size_t)hs.num_tris);
    std::memcpy(s->triv1.data(), hs.triv1, sizeof(float) * 4 * (size_t)hs.num_tris);
    std::memcpy(s->triv2.data(), hs.triv2, sizeof(float) * 4 * (size_t)hs.num_tris);
    s->nodeLinks.assign(hs.links, hs.links + 2 * (size_t)hs.num_nodes);
    s->triangleColours.resize((size_t)hs.num_tris);
    rt_host_make_colours(colourSeed, hs.num_tris, &s->triangleColours[0][0]);
    rt_host_free_scene(&hs);
    return s;
}
