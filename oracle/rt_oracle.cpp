// CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A line-by-line CPU restatement of the reference's render path, used only as the checker
// in tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline.  The shipped library
// (librt_hip.so) never links, loads or calls anything in this directory.
//
// Parity status: the reference holds no golden vectors, known-answer tests or fixtures for
// this path (it has no tests at all) and its GLSL cannot execute in the build container
// (no GL context can be created), so the restatement is pinned by: the integer RNG
// known-answers, the analytic hits of models/model.obj, the brute-force all-triangles
// cross-check the shader itself carries in a comment (raytrace.glsl:134-143), and the
// reference's own host objects compiled from source for the scene inputs.  GLSL leaves the
// precision of normalize/log/cos/pow/min/max and FMA contraction to the driver; this file
// fixes them through include/rt_math.h, which is therefore part of the oracle's definition.
//
// What each function follows:
//   tri_distance      shaders/raytrace.glsl:79-110   getRayTriangleDistance
//   box_distance      shaders/raytrace.glsl:113-125  rayBoundingBoxDist
//   closest_hit       shaders/raytrace.glsl:127-195  getHitInfo
//   brute_force_hit   shaders/raytrace.glsl:134-143  (commented-out all-triangles loop)
//   sample_sky        shaders/raytrace.glsl:202-211  sampleSkybox
//   rand/normal/dir   shaders/raytrace.glsl:213-239  rand .. randDirectionInHemisphere
//   path_colour       shaders/raytrace.glsl:241-270  getColour
//   shade_pixel       shaders/raytrace.glsl:272-305  main
//   oracle_normals    shaders/normals.glsl:13-19
//   buffer layouts    scene-loader.h:7-15, scene-loader.cpp:13-27, raytrace.cpp:38-57
//   uniforms          raytrace.cpp:65-73,80-85; constants.h:21-24
//
// Build: g++ -O2 -ffp-contract=off -fopenmp (see oracle/Makefile).

#include "rt_api.h"
#include "rt_math.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int kStackSize = 128;  // MAX_BVH_TRAVERSAL_STACK_SIZE, raytrace.glsl:8

struct Ray {
    rt_vec3 origin, dir, invDir;
};

struct HitInfo {
    float dist;
    int triangleIndex;
};

struct PixelCounters {  // the shader's per-invocation globals, raytrace.glsl:58-66
    int numBoxTests = 0;
    int numTriangleTests = 0;
    int numReflections = 0;
    // bookkeeping that is not in the shader
    unsigned long long rays = 0, hits = 0, pops = 0;
    int maxStack = 0;  // deepest stack index the traversal wrote to
};

struct Scene {
    int numNodes, numTris;
    const float* bvh;    // numNodes x 12
    const float* triv0;  // numTris x 4
    const float* triv1;
    const float* triv2;
    const float* colours;
    const float* normals;  // numTris x 4 (std430 vec3[] stride)
    const int32_t* links;  // optional exact (a, b) per node (rt_scene.links); NULL: the reference's float-encoded ids
};

inline rt_vec3 load3(const float* base, int i) { return rt_v3(base[4 * i], base[4 * i + 1], base[4 * i + 2]); }

float tri_distance(const Scene& sc, const Ray& ray, int triangleIndex, PixelCounters& pc) {
    pc.numTriangleTests++;
    rt_vec3 a = load3(sc.triv0, triangleIndex);
    rt_vec3 b = load3(sc.triv1, triangleIndex);
    rt_vec3 c = load3(sc.triv2, triangleIndex);

    rt_vec3 edge1 = rt_sub(b, a);
    rt_vec3 edge2 = rt_sub(c, a);
    rt_vec3 ray_cross_e2 = rt_cross(ray.dir, edge2);
    float det = rt_dot(edge1, ray_cross_e2);

    if (rt_abs(det) < RT_EPS) return RT_INF;

    float inv_det = 1.0f / det;
    rt_vec3 s = rt_sub(ray.origin, a);
    float u = inv_det * rt_dot(s, ray_cross_e2);

    if ((u < 0.0f && rt_abs(u) > RT_EPS) || (u > 1.0f && rt_abs(u - 1.0f) > RT_EPS)) return RT_INF;

    rt_vec3 s_cross_e1 = rt_cross(s, edge1);
    float v = inv_det * rt_dot(ray.dir, s_cross_e1);

    if ((v < 0.0f && rt_abs(v) > RT_EPS) || (u + v > 1.0f && rt_abs((u + v) - 1.0f) > RT_EPS)) return RT_INF;

    float t = inv_det * rt_dot(edge2, s_cross_e1);
    return t > RT_EPS ? t : RT_INF;
}

float box_distance(const Ray& ray, rt_vec3 boxMin, rt_vec3 boxMax, PixelCounters& pc) {
    pc.numBoxTests++;
    rt_vec3 tMin = rt_mul(rt_sub(boxMin, ray.origin), ray.invDir);
    rt_vec3 tMax = rt_mul(rt_sub(boxMax, ray.origin), ray.invDir);
    rt_vec3 t1 = rt_v3(rt_min(tMin.x, tMax.x), rt_min(tMin.y, tMax.y), rt_min(tMin.z, tMax.z));
    rt_vec3 t2 = rt_v3(rt_max(tMin.x, tMax.x), rt_max(tMin.y, tMax.y), rt_max(tMin.z, tMax.z));
    float tNear = rt_max(rt_max(t1.x, t1.y), t1.z);
    float tFar = rt_min(rt_min(t2.x, t2.y), t2.z);

    bool hit = tFar >= tNear && tFar > RT_EPS;
    float dst = hit ? (tNear > RT_EPS ? tNear : 0.0f) : RT_INF;
    return dst;
}

inline rt_vec3 node_min(const Scene& sc, int n) { return rt_v3(sc.bvh[12 * n], sc.bvh[12 * n + 1], sc.bvh[12 * n + 2]); }
inline rt_vec3 node_max(const Scene& sc, int n) {
    return rt_v3(sc.bvh[12 * n + 4], sc.bvh[12 * n + 5], sc.bvh[12 * n + 6]);
}

HitInfo closest_hit(const Scene& sc, const Ray& ray, PixelCounters& pc) {
    pc.rays++;
    HitInfo info;
    info.dist = RT_INF;
    info.triangleIndex = -1;
    if (box_distance(ray, node_min(sc, 0), node_max(sc, 0), pc) == RT_INF) return info;
    int stack[kStackSize];
    float dist[kStackSize];
    int i = 0;
    stack[0] = 0;
    dist[0] = box_distance(ray, node_min(sc, 0), node_max(sc, 0), pc);
    while (i > -1) {
        int bvhIndex = stack[i];
        if (dist[i--] >= info.dist) continue;
        pc.pops++;
        const float* node = sc.bvh + 12 * bvhIndex;
        bool isLeaf = rt_abs(node[8] - 1.0f) <= RT_EPS;
        if (isLeaf) {
            int triangleStart = sc.links ? sc.links[2 * (size_t)bvhIndex] : (int)node[9];
            int triangleEnd = sc.links ? sc.links[2 * (size_t)bvhIndex + 1] : (int)node[10];
            for (int k = triangleStart; k <= triangleEnd; k++) {
                float triangleDist = tri_distance(sc, ray, k, pc);
                if (triangleDist < info.dist) {
                    info.dist = triangleDist;
                    info.triangleIndex = k;
                }
            }
        } else {
            int child1 = sc.links ? sc.links[2 * (size_t)bvhIndex] : (int)node[9];
            int child2 = sc.links ? sc.links[2 * (size_t)bvhIndex + 1] : (int)node[10];
            float d1 = box_distance(ray, node_min(sc, child1), node_max(sc, child1), pc);
            float d2 = box_distance(ray, node_min(sc, child2), node_max(sc, child2), pc);
            if (d1 < d2) {
                if (d2 < info.dist) {
                    stack[++i] = child2;
                    dist[i] = d2;
                }
                if (d1 < info.dist) {
                    stack[++i] = child1;
                    dist[i] = d1;
                }
                if (i > pc.maxStack) pc.maxStack = i;
            } else {
                if (d1 < info.dist) {
                    stack[++i] = child1;
                    dist[i] = d1;
                }
                if (d2 < info.dist) {
                    stack[++i] = child2;
                    dist[i] = d2;
                }
                if (i > pc.maxStack) pc.maxStack = i;
            }
        }
    }
    if (info.triangleIndex >= 0) pc.hits++;
    return info;
}

HitInfo brute_force_hit(const Scene& sc, const Ray& ray, PixelCounters& pc) {
    HitInfo info;
    info.dist = RT_INF;
    info.triangleIndex = -1;
    for (int k = 0; k < sc.numTris; k++) {
        float triangleDist = tri_distance(sc, ray, k, pc);
        if (triangleDist < info.dist) {
            info.dist = triangleDist;
            info.triangleIndex = k;
        }
    }
    return info;
}

rt_vec3 sample_sky(rt_vec3 dir) {
    const rt_vec3 ground = rt_v3(0.6392156862f, 0.5803921f, 0.6392156862f);
    const rt_vec3 horizon = rt_v3(1.0f, 1.0f, 1.0f);
    const rt_vec3 zenith = rt_v3(0.486274509f, 0.71372549f, 234.0f / 255.0f);
    float skyGradientT = rt_pow(rt_smoothstep(0.0f, 0.4f, dir.y), 0.35f);
    float groundToSkyT = rt_smoothstep(-0.01f, 0.0f, dir.y);
    rt_vec3 g = rt_v3(rt_mix(horizon.x, zenith.x, skyGradientT), rt_mix(horizon.y, zenith.y, skyGradientT),
                      rt_mix(horizon.z, zenith.z, skyGradientT));
    return rt_v3(rt_mix(ground.x, g.x, groundToSkyT), rt_mix(ground.y, g.y, groundToSkyT),
                 rt_mix(ground.z, g.z, groundToSkyT));
}

float rand_normal(uint32_t* seed) {
    float theta = (2.0f * RT_PI) * rt_rand(seed);
    float rho = rt_sqrt(-2.0f * rt_log(rt_rand(seed)));
    return rho * rt_cos(theta);
}

rt_vec3 rand_direction(uint32_t* seed) {
    float x = rand_normal(seed);
    float y = rand_normal(seed);
    float z = rand_normal(seed);
    return rt_normalize(rt_v3(x, y, z));
}

rt_vec3 rand_direction_in_hemisphere(rt_vec3 normal, uint32_t* seed) {
    rt_vec3 d = rand_direction(seed);
    if (rt_dot(normal, d) < 0.0f) d = rt_neg(d);
    return d;
}

// returns rgb; alpha is the constant 1 of raytrace.glsl:269
rt_vec3 path_colour(const Scene& sc, Ray ray, uint32_t bouncesLeft, bool fixInvDir, uint32_t* seed,
                    PixelCounters& pc) {
    rt_vec3 rayColour = rt_v3(1.0f, 1.0f, 1.0f);
    rt_vec3 result = rt_v3(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < bouncesLeft; i++) {
        HitInfo info = closest_hit(sc, ray, pc);
        if (info.dist < RT_INF) {
            ray.origin = rt_add(ray.origin, rt_scale(ray.dir, info.dist));
            rt_vec3 normal = load3(sc.normals, info.triangleIndex);
            if (rt_dot(normal, ray.dir) > 0.0f) normal = rt_neg(normal);
            ray.dir = rand_direction_in_hemisphere(normal, seed);
            if (fixInvDir)  // not in the reference: RT_FLAG_FIX_INVDIR
                ray.invDir = rt_v3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
            rayColour = rt_mul(rayColour, load3(sc.colours, info.triangleIndex));
            if (i > 2) {
                float continueProb = 0.8f;
                if (rt_rand(seed) > continueProb) break;
                rayColour = rt_v3(rayColour.x / continueProb, rayColour.y / continueProb, rayColour.z / continueProb);
            }
        } else {
            result = rt_add(result, rt_mul(sample_sky(ray.dir), rayColour));
            break;
        }
        pc.numReflections++;
    }
    return result;
}

float pixel_width(const rt_camera* cam, int width) {
    // tan(u_FOV / 2.0f) * u_ViewportDist * 2.0f / u_ScreenWidth ; tan rounded correctly from double
    float tanHalf = (float)std::tan((double)(cam->fov_rad / 2.0f));
    return ((tanHalf * cam->viewport_dist) * 2.0f) / (float)width;
}

Ray primary_ray(const rt_camera* cam, float pixelWidth, int width, int height, int gx, int gy) {
    float W = (float)width, H = (float)height;
    rt_vec3 local = rt_v3(((float)gx - W / 2.0f) * pixelWidth, ((float)gy - H / 2.0f) * pixelWidth, cam->viewport_dist);
    Ray ray;
    ray.dir = rt_normalize(rt_mat3_mul(cam->rot, local));
    ray.origin = rt_v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    ray.invDir = rt_v3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
    return ray;
}

void shade_pixel(const Scene& sc, const rt_camera* cam, const rt_params* p, float pixelWidth, uint32_t frameCount,
                 int gx, int gy, const float* prev, float* out, PixelCounters& pc) {
    float W = (float)p->width;
    uint32_t pixelIndex = (uint32_t)((float)gy * W + (float)gx);
    uint32_t seed = pixelIndex + frameCount * 745621u;
    Ray ray = primary_ray(cam, pixelWidth, p->width, p->height, gx, gy);
    float colour[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    bool fix = (p->flags & RT_FLAG_FIX_INVDIR) != 0;
    for (uint32_t s = 0; s < p->rays_per_pixel; s++) {
        rt_vec3 c = path_colour(sc, ray, p->ray_bounces, fix, &seed, pc);
        colour[0] += c.x;
        colour[1] += c.y;
        colour[2] += c.z;
        colour[3] += 1.0f;
    }
    float rpp = (float)p->rays_per_pixel;
    for (int k = 0; k < 4; k++) colour[k] = colour[k] / rpp;
    size_t at = ((size_t)gy * (size_t)p->width + (size_t)gx) * 4;
    float fin[4];
    if (p->flags & RT_FLAG_SUM) {
        for (int k = 0; k < 4; k++) fin[k] = (prev ? prev[at + k] : 0.0f) + colour[k];
    } else {
        float fc = (float)frameCount, fc1 = (float)(frameCount + 1u);
        for (int k = 0; k < 4; k++) {
            float acc = (prev && frameCount != 0u) ? prev[at + k] : 0.0f;
            fin[k] = (acc * fc + colour[k]) / fc1;
        }
    }
    if (p->render_mode == RT_MODE_TRIANGLE_TESTS) {
        float f = (float)(pc.numTriangleTests < 500 ? pc.numTriangleTests : 500) / 500.0f;
        fin[0] = f * 1.0f; fin[1] = f * 1.0f; fin[2] = f * 0.0f; fin[3] = f * 0.0f;
    } else if (p->render_mode == RT_MODE_BOX_TESTS) {
        float f = (float)(pc.numBoxTests < 1000 ? pc.numBoxTests : 1000) / 1000.0f;
        fin[0] = f * 0.0f; fin[1] = f * 1.0f; fin[2] = f * 1.0f; fin[3] = f * 1.0f;
    } else if (p->render_mode == RT_MODE_REFLECTIONS) {
        uint32_t m = (uint32_t)pc.numReflections < p->ray_bounces ? (uint32_t)pc.numReflections : p->ray_bounces;
        float f = (float)m / (float)p->ray_bounces;
        fin[0] = f; fin[1] = f; fin[2] = f; fin[3] = f;
    }
    for (int k = 0; k < 4; k++) out[at + k] = fin[k];
}

Scene make_scene(const rt_scene* s, const float* normals) {
    Scene sc;
    sc.numNodes = s->num_nodes;
    sc.numTris = s->num_tris;
    sc.bvh = s->nodes;
    sc.triv0 = s->triv0;
    sc.triv1 = s->triv1;
    sc.triv2 = s->triv2;
    sc.colours = s->colours;
    sc.normals = normals;
    sc.links = s->links;
    return sc;
}

}  // namespace

extern "C" {

// shaders/normals.glsl:13-19.  out: num_tris x 4 floats (w = 0).
void oracle_normals(const rt_scene* s, float* out) {
    for (int i = 0; i < s->num_tris; i++) {
        rt_vec3 a = load3(s->triv0, i), b = load3(s->triv1, i), c = load3(s->triv2, i);
        rt_vec3 n = rt_normalize(rt_cross(rt_sub(a, b), rt_sub(a, c)));
        out[4 * i] = n.x;
        out[4 * i + 1] = n.y;
        out[4 * i + 2] = n.z;
        out[4 * i + 3] = 0.0f;
    }
}

// One dispatch (one frame).  prev may alias out.  counters (optional): RT_NUM_COUNTERS words, added to.
// rows [row_begin,row_end) only (0,height for everything) so that a bounded sample can be timed.
void oracle_frame_rows(const rt_scene* s, const float* normals, const rt_camera* cam, const rt_params* p,
                       uint32_t frame_count, const float* prev, float* out, unsigned long long* counters,
                       int row_begin, int row_end, int num_threads) {
    Scene sc = make_scene(s, normals);
    float pw = pixel_width(cam, p->width);
    unsigned long long box = 0, tri = 0, rays = 0, hits = 0, pops = 0, paths = 0;
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(+ : box, tri, rays, hits, pops, paths)
#endif
    for (int gy = row_begin; gy < row_end; gy++) {
        for (int gx = 0; gx < p->width; gx++) {
            PixelCounters pc;
            shade_pixel(sc, cam, p, pw, frame_count, gx, gy, prev, out, pc);
            box += (unsigned long long)pc.numBoxTests;
            tri += (unsigned long long)pc.numTriangleTests;
            rays += pc.rays;
            hits += pc.hits;
            pops += pc.pops;
            paths += p->rays_per_pixel;
        }
    }
    if (counters) {
        counters[0] += box;
        counters[1] += tri;
        counters[2] += rays;
        counters[3] += hits;
        counters[4] += pops;
        counters[5] += paths;
    }
}

// render(scene, camera, params) -> framebuffer, the frame loop of main.cpp:183-203 with the accumulation
// of raytrace.glsl:295-296 done in place.  frame_base == 0 starts from zeros.
void oracle_render(const rt_scene* s, const rt_camera* cam, const rt_params* p, float* rgba,
                   unsigned long long* counters, int num_threads) {
    std::vector<float> normals((size_t)s->num_tris * 4);
    oracle_normals(s, normals.data());
    uint32_t stride = p->frame_stride ? p->frame_stride : 1u;
    if (p->frame_base == 0) std::memset(rgba, 0, sizeof(float) * 4 * (size_t)p->width * (size_t)p->height);
    for (uint32_t f = 0; f < p->frames; f++) {
        uint32_t fc = p->frame_base + f * stride;
        oracle_frame_rows(s, normals.data(), cam, p, fc, rgba, rgba, counters, 0, p->height, num_threads);
    }
}

// Closest hit of every primary ray.  brute_force != 0 uses the all-triangles loop instead of the BVH.
void oracle_primary_hits(const rt_scene* s, const rt_camera* cam, int width, int height, int brute_force,
                         int32_t* ids, float* t, int num_threads) {
    Scene sc = make_scene(s, nullptr);
    float pw = pixel_width(cam, width);
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
#endif
    for (int gy = 0; gy < height; gy++) {
        for (int gx = 0; gx < width; gx++) {
            PixelCounters pc;
            Ray ray = primary_ray(cam, pw, width, height, gx, gy);
            HitInfo h = brute_force ? brute_force_hit(sc, ray, pc) : closest_hit(sc, ray, pc);
            ids[(size_t)gy * width + gx] = h.triangleIndex;
            t[(size_t)gy * width + gx] = h.dist;
        }
    }
}

// Per-pixel work statistics of one frame (analysis aid): number of rays of the pixel's paths, node pops of
// its most expensive ray, and total node pops.
void oracle_path_stats(const rt_scene* s, const float* normals, const rt_camera* cam, const rt_params* p,
                       uint32_t frame_count, uint32_t* out_rays, uint32_t* out_max_pops, uint32_t* out_pops,
                       uint32_t* out_max_stack, int num_threads) {
    Scene sc = make_scene(s, normals);
    float pw = pixel_width(cam, p->width);
    bool fix = (p->flags & RT_FLAG_FIX_INVDIR) != 0;
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
#endif
    for (int gy = 0; gy < p->height; gy++) {
        for (int gx = 0; gx < p->width; gx++) {
            uint32_t pixelIndex = (uint32_t)((float)gy * (float)p->width + (float)gx);
            uint32_t seed = pixelIndex + frame_count * 745621u;
            Ray ray = primary_ray(cam, pw, p->width, p->height, gx, gy);
            rt_vec3 thr = rt_v3(1, 1, 1);
            uint32_t rays = 0, maxp = 0, pops = 0;
            for (uint32_t i = 0; i < p->ray_bounces; i++) {
                PixelCounters pc;
                HitInfo info = closest_hit(sc, ray, pc);
                rays++;
                pops += (uint32_t)pc.pops;
                if (pc.pops > maxp) maxp = (uint32_t)pc.pops;
                if (out_max_stack && (uint32_t)pc.maxStack > out_max_stack[(size_t)gy * p->width + gx])
                    out_max_stack[(size_t)gy * p->width + gx] = (uint32_t)pc.maxStack;
                if (!(info.dist < RT_INF)) break;
                ray.origin = rt_add(ray.origin, rt_scale(ray.dir, info.dist));
                rt_vec3 normal = load3(sc.normals, info.triangleIndex);
                if (rt_dot(normal, ray.dir) > 0.0f) normal = rt_neg(normal);
                ray.dir = rand_direction_in_hemisphere(normal, &seed);
                if (fix) ray.invDir = rt_v3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
                if (i > 2) {
                    if (rt_rand(&seed) > 0.8f) break;
                }
            }
            (void)thr;
            size_t at = (size_t)gy * p->width + gx;
            out_rays[at] = rays;
            out_max_pops[at] = maxp;
            out_pops[at] = pops;
        }
    }
}

// ---- unit hooks for known-answer tests -------------------------------------------------------
void oracle_rand(uint32_t seed, int n, uint32_t* raw, float* as_float) {
    for (int i = 0; i < n; i++) {
        uint32_t s2 = seed;
        raw[i] = rt_rand_raw(&s2);
        as_float[i] = rt_rand(&seed);
    }
}

float oracle_tri_distance(const float origin[3], const float dir[3], const float a[3], const float b[3],
                          const float c[3]) {
    float v0[4] = {a[0], a[1], a[2], 0}, v1[4] = {b[0], b[1], b[2], 0}, v2[4] = {c[0], c[1], c[2], 0};
    Scene sc{};
    sc.triv0 = v0;
    sc.triv1 = v1;
    sc.triv2 = v2;
    Ray r;
    r.origin = rt_v3(origin[0], origin[1], origin[2]);
    r.dir = rt_v3(dir[0], dir[1], dir[2]);
    r.invDir = rt_v3(0, 0, 0);
    PixelCounters pc;
    return tri_distance(sc, r, 0, pc);
}

float oracle_box_distance(const float origin[3], const float inv_dir[3], const float bmin[3], const float bmax[3]) {
    Ray r;
    r.origin = rt_v3(origin[0], origin[1], origin[2]);
    r.dir = rt_v3(0, 0, 0);
    r.invDir = rt_v3(inv_dir[0], inv_dir[1], inv_dir[2]);
    PixelCounters pc;
    return box_distance(r, rt_v3(bmin[0], bmin[1], bmin[2]), rt_v3(bmax[0], bmax[1], bmax[2]), pc);
}

void oracle_sky(const float dir[3], float out[3]) {
    rt_vec3 c = sample_sky(rt_v3(dir[0], dir[1], dir[2]));
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
}

void oracle_hemisphere_dir(uint32_t* seed, const float normal[3], float out[3]) {
    rt_vec3 d = rand_direction_in_hemisphere(rt_v3(normal[0], normal[1], normal[2]), seed);
    out[0] = d.x;
    out[1] = d.y;
    out[2] = d.z;
}

void oracle_primary_ray(const rt_camera* cam, int width, int height, int gx, int gy, float out_dir[3],
                        float out_inv[3]) {
    Ray r = primary_ray(cam, pixel_width(cam, width), width, height, gx, gy);
    out_dir[0] = r.dir.x; out_dir[1] = r.dir.y; out_dir[2] = r.dir.z;
    out_inv[0] = r.invDir.x; out_inv[1] = r.invDir.y; out_inv[2] = r.invDir.z;
}

// rt_math.h on the host, same output layout as rt_selftest_math (rt_api.h)
void oracle_math(const float* x, int n, float* out) {
    for (int i = 0; i < n; i++) {
        out[i] = rt_log(x[i]);
        out[n + i] = rt_cos(x[i]);
        out[2 * n + i] = rt_exp(x[i]);
        out[3 * n + i] = rt_sqrt(x[i]);
        out[4 * n + i] = 1.0f / x[i];
        out[5 * n + i] = rt_pow(x[i], 0.35f);
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
