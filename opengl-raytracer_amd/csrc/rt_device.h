// Device-side building blocks of the render path (gfx950).  Every floating-point operation goes
// through include/rt_math.h so that results are bit-identical to the CPU oracle; this file decides
// only data layout, control flow and where state lives (VGPR / LDS / HBM).
//
// Reference functions realised here (shaders/raytrace.glsl):
//   box_dist        :113-125 rayBoundingBoxDist     tri_dist      :79-110  getRayTriangleDistance
//   closest_hit     :127-195 getHitInfo             sample_sky    :202-211 sampleSkybox
//   hemisphere_dir  :221-239 randNormalDistribution..randDirectionInHemisphere
//   bounce          :246-262 the hit branch of getColour
//   primary_ray     :276-288 seed + primary ray of main
//
// Scene layout in HBM (built by rt_upload_scene from the reference's buffers):
//   nodes  one 64-byte record per INNER node, 64-byte aligned, holding BOTH children's boxes and
//          references, so one pop costs one 64-byte fetch instead of the reference's three 48-byte
//          mat3 reads (self + two children, raytrace.glsl:154,169-170):
//            q0 = (c1.min.x c1.min.y c1.min.z c1.max.x)   q1 = (c1.max.y c1.max.z c2.min.x c2.min.y)
//            q2 = (c2.min.z c2.max.x c2.max.y c2.max.z)   q3 = (ref1 ref2 - -) as int bits
//          ref >= 0: inner record index; ref < 0: leaf, first triangle record = ~ref;
//          ref == RT_REF_EMPTY: leaf with an empty range.
//   tris   one 64-byte record per triangle (48 used; padded so a record never straddles a cache line),
//          in leaf order:
//            q0 = (a.xyz, flags) flags bit0 = last triangle of its leaf
//            q1 = (b-a, id)      id = index in the caller's triangle arrays (what hits report)
//            q2 = (c-a, 0)       the edges are the reference's own first two subtractions (:85-86)
//   shade  two float4 per caller triangle id: normal (normals.glsl) and colour (rgb).
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_math.h"

#define RT_REF_IDLE ((int)0x80000000)    // "no node": what an idle lane of the persistent kernel holds
#define RT_REF_EMPTY ((int)0x80000001)   // leaf with an empty triangle range
#define RT_BLOCK 256          // threads per workgroup of every kernel here (4 waves)
#define RT_LDS_STACK 8        // traversal-stack entries per lane kept in LDS; deeper ones spill to HBM

struct DScene {
    const float4* nodes;
    const float4* tris;
    const float4* shade;
    float rmin[3];
    float rmax[3];
    int root_ref;
    int num_tris;
};

struct DCamera {
    float pos[3];
    float rot[9];
    float viewport_dist;
    float pixel_width;   // tan(fov/2) * viewport_dist * 2 / width, computed once on the host
    float half_w, half_h;  // width / 2.0f, height / 2.0f
    float width_f;
    int width, height;
};

struct Counters {
    uint32_t box, tri, rays, hits, pops;
};

// Per-lane traversal stack: entries [0, RT_LDS_STACK) in LDS as (ref, dist) pairs laid out
// [entry][thread] (conflict-free for 8-byte accesses), deeper entries in a global spill area
// laid out [entry][slot] so that lanes of a wave touch neighbouring addresses.
struct Stack {
    uint2* lds;        // &s_stack[0][threadIdx.x]
    uint2* spill;      // &spill[0][slot] or nullptr when the scene needs no spill
    size_t spill_stride;
    __device__ __forceinline__ void put(int i, int ref, float dist) {
        uint2 v = make_uint2((uint32_t)ref, rt_f2u(dist));
        if (i < RT_LDS_STACK)
            lds[i * RT_BLOCK] = v;
        else
            spill[(size_t)(i - RT_LDS_STACK) * spill_stride] = v;
    }
    __device__ __forceinline__ uint2 get(int i) const {
        if (i < RT_LDS_STACK) return lds[i * RT_BLOCK];
        return spill[(size_t)(i - RT_LDS_STACK) * spill_stride];
    }
};

__device__ __forceinline__ float box_dist(rt_vec3 o, rt_vec3 inv, float nx, float ny, float nz, float xx, float xy,
                                          float xz) {
    float ax = (nx - o.x) * inv.x, ay = (ny - o.y) * inv.y, az = (nz - o.z) * inv.z;
    float bx = (xx - o.x) * inv.x, by = (xy - o.y) * inv.y, bz = (xz - o.z) * inv.z;
    float tNear = rt_max(rt_max(rt_min(ax, bx), rt_min(ay, by)), rt_min(az, bz));
    float tFar = rt_min(rt_min(rt_max(ax, bx), rt_max(ay, by)), rt_max(az, bz));
    bool hit = tFar >= tNear && tFar > RT_EPS;
    return hit ? (tNear > RT_EPS ? tNear : 0.0f) : RT_INF;
}

__device__ __forceinline__ float tri_dist(rt_vec3 o, rt_vec3 d, rt_vec3 a, rt_vec3 e1, rt_vec3 e2) {
    rt_vec3 p = rt_cross(d, e2);
    float det = rt_dot(e1, p);
    if (rt_abs(det) < RT_EPS) return RT_INF;
    float inv_det = 1.0f / det;
    rt_vec3 s = rt_sub(o, a);
    float u = inv_det * rt_dot(s, p);
    if ((u < 0.0f && rt_abs(u) > RT_EPS) || (u > 1.0f && rt_abs(u - 1.0f) > RT_EPS)) return RT_INF;
    rt_vec3 q = rt_cross(s, e1);
    float v = inv_det * rt_dot(d, q);
    if ((v < 0.0f && rt_abs(v) > RT_EPS) || (u + v > 1.0f && rt_abs((u + v) - 1.0f) > RT_EPS)) return RT_INF;
    float t = inv_det * rt_dot(e2, q);
    return t > RT_EPS ? t : RT_INF;
}

// The same function without early exits: every lane computes u, v and t and the rejections are folded into one select.
// Identical results (the rejected lanes' values are never used, the tests are the same expressions); for a wave whose
// lanes leave at different points this is cheaper than one masked branch per exit, and all three loads of the triangle
// record are issued together.
__device__ __forceinline__ float tri_dist_flat(rt_vec3 o, rt_vec3 d, rt_vec3 a, rt_vec3 e1, rt_vec3 e2) {
    rt_vec3 p = rt_cross(d, e2);
    float det = rt_dot(e1, p);
    float inv_det = 1.0f / det;
    rt_vec3 s = rt_sub(o, a);
    float u = inv_det * rt_dot(s, p);
    rt_vec3 q = rt_cross(s, e1);
    float v = inv_det * rt_dot(d, q);
    float t = inv_det * rt_dot(e2, q);
    int rej = (int)(rt_abs(det) < RT_EPS);
    rej |= (int)(u < 0.0f) & (int)(rt_abs(u) > RT_EPS);
    rej |= (int)(u > 1.0f) & (int)(rt_abs(u - 1.0f) > RT_EPS);
    rej |= (int)(v < 0.0f) & (int)(rt_abs(v) > RT_EPS);
    rej |= (int)(u + v > 1.0f) & (int)(rt_abs((u + v) - 1.0f) > RT_EPS);
    rej |= (int)!(t > RT_EPS);
    return rej ? RT_INF : t;
}

// Ordered depth-first closest hit.  Visit order, the `dist >= best` cull at pop, the strict `<`
// updates and the tie rule (d1 == d2 visits child 2 first) are the reference's; the nearer child is
// continued with directly instead of being pushed and popped again, which is the same sequence.
// best_id is the caller's triangle id or -1.
//
// All lanes that enter stay in the loop until the last one has finished: every iteration the wave runs
// ONE kind of step -- inner-node visit or triangle test -- whichever more of its lanes wait for, under a
// wave-uniform branch, instead of executing both bodies with complementary masks.
template <bool COUNT>
__device__ __forceinline__ void closest_hit(const DScene& sc, rt_vec3 o, rt_vec3 d, rt_vec3 inv, Stack st, float& best,
                                            int& best_id, Counters& cn) {
    best = RT_INF;
    best_id = -1;
    if (COUNT) {
        cn.rays++;
        cn.box++;
    }
    float d0 = box_dist(o, inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2]);
    int cur = RT_REF_IDLE;  // no node: the lane has finished
    int sp = 0;             // entries currently on the stack
    if (d0 != RT_INF) {
        if (COUNT) cn.box++;  // the reference evaluates the root box a second time to seed the stack
        cur = sc.root_ref;
        if (COUNT && cur < 0) cn.pops++;
    }
#define RT_POP()                                \
    for (;;) {                                  \
        if (sp == 0) {                          \
            cur = RT_REF_IDLE;                  \
            break;                              \
        }                                       \
        sp--;                                   \
        uint2 e_ = st.get(sp);                  \
        if (!(rt_u2f(e_.y) >= best)) {          \
            cur = (int)e_.x;                    \
            if (COUNT && cur < 0) cn.pops++;    \
            break;                              \
        }                                       \
    }
    for (;;) {
        const bool want_inner = cur >= 0;
        const bool want_leaf = (uint32_t)cur > 0x80000000u;
        const uint32_t ni = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want_inner));
        const uint32_t nl = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want_leaf));
        if (ni + nl == 0u) break;
        if (ni >= nl) {
            if (want_inner) {
                const float4* rec = sc.nodes + (size_t)cur * 4;
                float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                if (COUNT) {
                    cn.pops++;
                    cn.box += 2;
                }
                float d1 = box_dist(o, inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                float d2 = box_dist(o, inv, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                int r1 = (int)rt_f2u(q3.x), r2 = (int)rt_f2u(q3.y);
                bool first_is_1 = d1 < d2;
                int rn = first_is_1 ? r1 : r2, rf = first_is_1 ? r2 : r1;
                float dn = first_is_1 ? d1 : d2, df = first_is_1 ? d2 : d1;
                if (df < best) {
                    st.put(sp, rf, df);
                    sp++;
                }
                if (dn < best) {
                    cur = rn;
                    if (COUNT && cur < 0) cn.pops++;
                } else {
                    RT_POP();
                }
            }
        } else {
            if (want_leaf) {
                bool last = true;
                if (cur != RT_REF_EMPTY) {
                    const int k = ~cur;
                    const float4* tr = sc.tris + (size_t)k * 4;
                    float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
                    if (COUNT) cn.tri++;
                    float t = tri_dist(o, d, rt_v3(t0.x, t0.y, t0.z), rt_v3(t1.x, t1.y, t1.z), rt_v3(t2.x, t2.y, t2.z));
                    if (t < best) {
                        best = t;
                        best_id = (int)rt_f2u(t1.w);
                    }
                    last = (rt_f2u(t0.w) & 1u) != 0u;
                    cur = ~(k + 1);
                }
                if (last) {
                    RT_POP();
                }
            }
        }
    }
#undef RT_POP
    if (COUNT && best_id >= 0) cn.hits++;
}

__device__ __forceinline__ rt_vec3 sample_sky(rt_vec3 dir) {
    float tSky = rt_pow(rt_smoothstep(0.0f, 0.4f, dir.y), 0.35f);
    float tGround = rt_smoothstep(-0.01f, 0.0f, dir.y);
    const float zx = 0.486274509f, zy = 0.71372549f, zz = 234.0f / 255.0f;
    float gx = rt_mix(1.0f, zx, tSky), gy = rt_mix(1.0f, zy, tSky), gz = rt_mix(1.0f, zz, tSky);
    return rt_v3(rt_mix(0.6392156862f, gx, tGround), rt_mix(0.5803921f, gy, tGround), rt_mix(0.6392156862f, gz, tGround));
}

__device__ __forceinline__ float rand_normal(uint32_t* seed) {
    float theta = (2.0f * RT_PI) * rt_rand(seed);
    float rho = rt_sqrt(-2.0f * rt_log(rt_rand(seed)));
    return rho * rt_cos(theta);
}

__device__ __forceinline__ rt_vec3 hemisphere_dir(rt_vec3 n, uint32_t* seed) {
    float x = rand_normal(seed);
    float y = rand_normal(seed);
    float z = rand_normal(seed);
    rt_vec3 r = rt_normalize(rt_v3(x, y, z));
    if (rt_dot(n, r) < 0.0f) r = rt_neg(r);
    return r;
}

// State of one path between two rays.
struct PathState {
    rt_vec3 o, d, inv, thr;
    uint32_t seed;
};

// The hit branch of getColour for bounce index i.  Returns false when Russian roulette ends the path.
template <bool FIX_INVDIR>
__device__ __forceinline__ bool bounce(const DScene& sc, PathState& p, float t, int id, uint32_t i) {
    p.o = rt_add(p.o, rt_scale(p.d, t));
    float4 nq = sc.shade[(size_t)id * 2], cq = sc.shade[(size_t)id * 2 + 1];
    rt_vec3 n = rt_v3(nq.x, nq.y, nq.z);
    if (rt_dot(n, p.d) > 0.0f) n = rt_neg(n);
    p.d = hemisphere_dir(n, &p.seed);
    if (FIX_INVDIR) p.inv = rt_v3(1.0f / p.d.x, 1.0f / p.d.y, 1.0f / p.d.z);
    p.thr = rt_mul(p.thr, rt_v3(cq.x, cq.y, cq.z));
    if (i > 2u) {
        const float continueProb = 0.8f;
        if (rt_rand(&p.seed) > continueProb) return false;
        p.thr = rt_v3(p.thr.x / continueProb, p.thr.y / continueProb, p.thr.z / continueProb);
    }
    return true;
}

__device__ __forceinline__ uint32_t pixel_seed(const DCamera& c, int gx, int gy, uint32_t frame) {
    uint32_t pixelIndex = (uint32_t)((float)gy * c.width_f + (float)gx);
    return pixelIndex + frame * 745621u;
}

__device__ __forceinline__ void primary_ray(const DCamera& c, int gx, int gy, rt_vec3& o, rt_vec3& d, rt_vec3& inv) {
    rt_vec3 local = rt_v3(((float)gx - c.half_w) * c.pixel_width, ((float)gy - c.half_h) * c.pixel_width, c.viewport_dist);
    d = rt_normalize(rt_mat3_mul(c.rot, local));
    o = rt_v3(c.pos[0], c.pos[1], c.pos[2]);
    inv = rt_v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

// Pixel of lane `tid` of 16x16 tile `tile`: each wave covers an 8x8 pixel square so that the
// 64 primary rays of a wave are neighbours in both directions.
__device__ __forceinline__ void tile_pixel(int tile, int tiles_x, int tid, int& gx, int& gy) {
    int wave = tid >> 6, lane = tid & 63;
    int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    gx = tx * 16 + (wave & 1) * 8 + (lane & 7);
    gy = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
}

// Running mean / sum of raytrace.glsl:296 for one channel.
__device__ __forceinline__ float accumulate(float acc, float colour, float fc, float fc1, bool sum_mode) {
    return sum_mode ? acc + colour : (acc * fc + colour) / fc1;
}

#endif  // RT_DEVICE_H
