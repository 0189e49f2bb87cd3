// C++ mirror of the reference's raytrace.h:9-10 above the C ABI of include/rt_api.h.
//
//   reference                                               here
//   GLuint raytraceInit(SceneData*, int w, int h)           Raytracer* raytraceInit(SceneData*, int w, int h, int device = 0)
//   void raytrace(cameraPos, cameraRotation, renderMode,    void raytrace(Raytracer*, cameraPos, cameraRotation, renderMode,
//                 frameCount, num_groups_x, num_groups_y)                 frameCount, prevFrame, outFrame)
//
// The reference's caller binds image unit 6 (previous frame, read) and 5 (current frame, written) before
// each call (main.cpp:189-192); here the two DEVICE buffers are explicit arguments.  The group counts
// disappear (the library chooses its own launch geometry).  Errors throw std::runtime_error with the
// library's message instead of being printed and ignored (util.cpp:9-14).
#pragma once
#include <array>

#include "rt_api.h"
#include "scene_loader.h"

using Vec3 = std::array<float, 3>;
using Mat3 = std::array<float, 9>;  // column-major, as glm::mat3 / glUniformMatrix3fv(.., GL_FALSE, ..)

// constants.h:21-30
constexpr float FOV = 90.0f;
constexpr float VIEWPORT_DIST = 0.1f;
constexpr unsigned RAY_BOUNCES = 100;
constexpr unsigned RAYS_PER_PIXEL = 1;
constexpr float CAMERA_MOVE_SPEED = 2.0f;
constexpr Vec3 CAMERA_START_POS = {0.0f, 0.0f, -2.0f};

struct Raytracer {
    rt_context* ctx = nullptr;
    int width = 0, height = 0;
    unsigned flags = 0;  // RT_FLAG_* applied to every dispatch
};

Raytracer* raytraceInit(SceneData* sceneData, int screenWidth, int screenHeight, int device = 0);
void raytrace(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int renderMode, int frameCount,
              const float* d_prevFrame, float* d_outFrame);
// render(scene, camera, spp) -> framebuffer: `frames` dispatches from frameBase accumulated in d_accum.
void raytraceAccumulate(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int frameBase, int frames, float* d_accum);
// One GPU's share of a partitioned accumulation (no counterpart in the reference, which knows one GPU):
//   samples  frames frameBase, frameBase + frameStride, ... summed into d_accum (RT_FLAG_SUM); the ranks' buffers are
//            added (ncclReduce, or a file combiner) and divided by the total frame count;
//   tiles    bandCount > 0: the running mean of consecutive frames, restricted to the interleaved band bandIndex of
//            bandCount (stripes of 16 pixel rows dealt round-robin); the ranks' rows are gathered.
void raytraceAccumulatePart(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int frameBase, int frameStride, int frames,
                            bool sum, int bandIndex, int bandCount, float* d_accum);
void raytraceShutdown(Raytracer* rt);
