#include "raytrace.h"

#include <cmath>
#include <stdexcept>
#include <string>

static void check(Raytracer* rt, int rc, const char* what) {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(rt ? rt->ctx : nullptr));
}

static rt_camera make_camera(Vec3 pos, Mat3 rot) {
    rt_camera cam;
    for (int k = 0; k < 3; k++) cam.pos[k] = pos[k];
    for (int k = 0; k < 9; k++) cam.rot[k] = rot[k];
    cam.fov_rad = static_cast<float>(FOV * 3.14159265358979323846 / 180.f);  // raytrace.cpp:69-70
    cam.viewport_dist = VIEWPORT_DIST;
    return cam;
}

static rt_params make_params(const Raytracer* rt, int renderMode, int frameBase, int frames) {
    rt_params p{};
    p.width = rt->width;
    p.height = rt->height;
    p.rays_per_pixel = RAYS_PER_PIXEL;
    p.ray_bounces = RAY_BOUNCES;
    p.render_mode = (uint32_t)renderMode;
    p.frame_base = (uint32_t)frameBase;
    p.frames = (uint32_t)frames;
    p.frame_stride = 1;
    p.flags = rt->flags;
    return p;
}

Raytracer* raytraceInit(SceneData* sceneData, int screenWidth, int screenHeight, int device) {
    auto* rt = new Raytracer();
    rt->width = screenWidth;
    rt->height = screenHeight;
    int rc = rt_create(device, &rt->ctx);
    if (rc != RT_OK) {
        std::string msg = rt_last_error(nullptr);
        delete rt;
        throw std::runtime_error("rt_create: " + msg);
    }
    rt_scene s{};
    s.num_nodes = (int)sceneData->alignedBVHNodes.size();
    s.num_tris = sceneData->numTriangles;
    s.nodes = &sceneData->alignedBVHNodes[0].u[0];
    s.triv0 = &sceneData->triv0[0][0];
    s.triv1 = &sceneData->triv1[0][0];
    s.triv2 = &sceneData->triv2[0][0];
    s.colours = &sceneData->triangleColours[0][0];
    s.links = sceneData->nodeLinks.empty() ? nullptr : sceneData->nodeLinks.data();
    check(rt, rt_upload_scene(rt->ctx, &s), "rt_upload_scene");
    return rt;
}

void raytrace(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int renderMode, int frameCount,
              const float* d_prevFrame, float* d_outFrame) {
    rt_camera cam = make_camera(cameraPos, cameraRotation);
    rt_params p = make_params(rt, renderMode, frameCount, 1);
    check(rt, rt_raytrace_frame(rt->ctx, &cam, &p, (uint32_t)frameCount, d_prevFrame, d_outFrame), "rt_raytrace_frame");
}

void raytraceAccumulate(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int frameBase, int frames, float* d_accum) {
    rt_camera cam = make_camera(cameraPos, cameraRotation);
    rt_params p = make_params(rt, RT_MODE_RENDER, frameBase, frames);
    check(rt, rt_render_device(rt->ctx, &cam, &p, d_accum), "rt_render_device");
}

void raytraceAccumulatePart(Raytracer* rt, Vec3 cameraPos, Mat3 cameraRotation, int frameBase, int frameStride, int frames,
                            bool sum, int bandIndex, int bandCount, float* d_accum) {
    rt_camera cam = make_camera(cameraPos, cameraRotation);
    rt_params p = make_params(rt, RT_MODE_RENDER, frameBase, frames);
    p.frame_stride = (uint32_t)frameStride;
    if (sum) p.flags |= RT_FLAG_SUM;
    if (bandCount > 0) p.band = ((uint32_t)bandIndex << 16) | (uint32_t)bandCount | RT_BAND_INTERLEAVED;
    check(rt, rt_render_device(rt->ctx, &cam, &p, d_accum), "rt_render_device");
}

void raytraceShutdown(Raytracer* rt) {
    if (!rt) return;
    rt_destroy(rt->ctx);
    delete rt;
}
