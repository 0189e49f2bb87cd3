// C++ mirror of the reference's scene-loader.h:7-17 above librt_host.so: the same SceneData the
// reference hands to raytraceInit, with std::vector storage (no GLM), plus the triangle colours that
// the reference generates inside initBuffers (raytrace.cpp:44-52) made explicit.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

struct AlignedMat3 {
    std::array<float, 4> u, v, w;  // {min.xyz,0} {max.xyz,0} {isLeaf,a,b,0}
};

struct SceneData {
    int numTriangles = 0;
    std::vector<AlignedMat3> alignedBVHNodes;
    std::vector<std::array<float, 4>> triv0, triv1, triv2;
    std::vector<std::array<float, 4>> triangleColours;  // rgb + 1
    std::vector<int32_t> nodeLinks;  // not in the reference: (a, b) of every node as exact integers (rt_scene.links)
};

// loadScene() for an explicit path (the reference compiles "../models/teapot.obj" in, constants.h:14).
// Throws std::runtime_error when the file cannot be opened, as the reference does (obj-reader.cpp:15).
SceneData* loadScene(const std::string& objPath, unsigned colourSeed = 1);
