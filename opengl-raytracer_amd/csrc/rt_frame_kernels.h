// Per-frame pipelines of the render path and the small utility kernels.
//
//   wavefront (RT_FLAG_WAVEFRONT)  generate -> [extend -> shade(+compact)] x RT_WF_BOUNCES -> tail
//       generate   one primary ray per pixel into a ray queue, 8x8 pixel squares per wave
//       extend     closest hit of every ray of a queue (BVH traversal + triangle tests)
//       shade      consumes the hits: sky + framebuffer accumulation for finished paths, diffuse
//                  bounce + Russian roulette for the others, survivors compacted into the next queue
//                  with a wave ballot, an mbcnt prefix and ONE atomic per wave
//       tail       the few paths still alive after RT_WF_BOUNCES bounces are finished by one launch
//                  in which each lane loops extend+shade, so the launch count per frame is fixed and
//                  the host never reads a queue size back
//   megakernel (RT_FLAG_MEGAKERNEL, debug render modes, rays_per_pixel > 1, ray_bounces == 0)
//       one lane follows one pixel's whole sample loop, the reference's dispatch shape
//       (raytrace.glsl:272-305).
#ifndef RT_FRAME_KERNELS_H
#define RT_FRAME_KERNELS_H

#include "rt_device.h"

#define RT_WF_BOUNCES 2       // bounces handled by wavefront kernels before the tail launch
#define RT_TAIL_BLOCKS 1024   // grid of the tail kernel (grid-stride)

// ------------------------------------------------------------------------------------------------
// device structures
// ------------------------------------------------------------------------------------------------
struct DQueue {
    float4* o;    // origin.xyz, w = pixel index bits (0xffffffff: slot holds no ray)
    float4* d;    // dir.xyz,    w = RNG seed bits
    float4* inv;  // invDir.xyz (the primary ray's unless RT_FLAG_FIX_INVDIR)
    float4* thr;  // throughput rgb
};

struct FrameArgs {
    uint32_t frame;       // frameCount of this dispatch
    uint32_t bounces;     // u_RayBounces
    uint32_t rpp;         // u_RaysPerPixel
    uint32_t mode;        // renderMode
    uint32_t sum_mode;    // RT_FLAG_SUM
    uint32_t read_prev;   // 0: treat the previous frame as zeros
    int tiles_x, num_tiles;
};

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// shaders/normals.glsl:13-19, one lane per triangle; also stores the colour next to the normal.
__global__ __launch_bounds__(RT_BLOCK) void k_normals(const float4* v0, const float4* v1, const float4* v2,
                                                      const float4* colours, float4* shade, int n) {
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 a = v0[i], b = v1[i], c = v2[i];
    rt_vec3 A = rt_v3(a.x, a.y, a.z), B = rt_v3(b.x, b.y, b.z), Cc = rt_v3(c.x, c.y, c.z);
    rt_vec3 nrm = rt_normalize(rt_cross(rt_sub(A, B), rt_sub(A, Cc)));
    shade[2 * (size_t)i] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
    float4 col = colours[i];
    shade[2 * (size_t)i + 1] = make_float4(col.x, col.y, col.z, 0.0f);
}

__device__ __forceinline__ void flush_counters(const Counters& cn, unsigned long long paths, unsigned long long* g) {
    // per-wave sums, then one atomic per counter per wave
    unsigned long long v[RT_NUM_COUNTERS] = {cn.box, cn.tri, cn.rays, cn.hits, cn.pops, paths};
#pragma unroll
    for (int k = 0; k < RT_NUM_COUNTERS; k++) {
        unsigned long long x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g[k], x);
    }
}

__device__ __forceinline__ Stack make_stack(uint2 (*s_stack)[RT_BLOCK], uint2* spill, size_t spill_stride, size_t slot) {
    Stack st;
    st.lds = &s_stack[0][threadIdx.x];
    st.spill = spill ? spill + slot : nullptr;
    st.spill_stride = spill_stride;
    return st;
}

__device__ __forceinline__ void store_pixel(const FrameArgs& fa, const float4* prev, float4* out, size_t at, float r,
                                            float g, float b, float a) {
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (fa.read_prev) acc = prev[at];
    float fc = (float)fa.frame, fc1 = (float)(fa.frame + 1u);
    bool s = fa.sum_mode != 0;
    out[at] = make_float4(accumulate(acc.x, r, fc, fc1, s), accumulate(acc.y, g, fc, fc1, s),
                          accumulate(acc.z, b, fc, fc1, s), accumulate(acc.w, a, fc, fc1, s));
}

// ---- megakernel ----------------------------------------------------------------------------------
template <bool COUNT, bool FIX>
__global__ __launch_bounds__(RT_BLOCK) void k_mega(DScene sc, DCamera cam, FrameArgs fa, const float4* prev, float4* out,
                                                   unsigned long long* g_counters, uint2* spill, size_t spill_stride) {
    __shared__ uint2 s_stack[RT_LDS_STACK][RT_BLOCK];
    int gx, gy;
    tile_pixel(blockIdx.x, fa.tiles_x, threadIdx.x, gx, gy);
    bool active = gx < cam.width && gy < cam.height;
    Counters cn = {0, 0, 0, 0, 0};
    unsigned long long paths = 0;
    if (active) {
        Stack st = make_stack(s_stack, spill, spill_stride, (size_t)blockIdx.x * RT_BLOCK + threadIdx.x);
        uint32_t reflections = 0;
        rt_vec3 o0, d0, inv0;
        primary_ray(cam, gx, gy, o0, d0, inv0);
        uint32_t seed = pixel_seed(cam, gx, gy, fa.frame);
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 1.0f;
        for (uint32_t s = 0; s < fa.rpp; s++) {
            PathState p;
            p.o = o0;
            p.d = d0;
            p.inv = inv0;
            p.thr = rt_v3(1.0f, 1.0f, 1.0f);
            p.seed = seed;
            rt_vec3 result = rt_v3(0.0f, 0.0f, 0.0f);
            for (uint32_t i = 0; i < fa.bounces; i++) {
                float t;
                int id;
                closest_hit<COUNT>(sc, p.o, p.d, p.inv, st, t, id, cn);
                if (t < RT_INF) {
                    if (!bounce<FIX>(sc, p, t, id, i)) break;
                } else {
                    result = rt_add(result, rt_mul(sample_sky(p.d), p.thr));
                    break;
                }
                reflections++;
            }
            seed = p.seed;
            cr += result.x;
            cg += result.y;
            cb += result.z;
            ca += 1.0f;
            paths++;
        }
        float rpp = (float)fa.rpp;
        cr = cr / rpp;
        cg = cg / rpp;
        cb = cb / rpp;
        ca = ca / rpp;
        size_t at = (size_t)gy * (size_t)cam.width + (size_t)gx;
        if (fa.mode == RT_MODE_RENDER || !COUNT) {
            store_pixel(fa, prev, out, at, cr, cg, cb, ca);
        } else if (fa.mode == RT_MODE_TRIANGLE_TESTS) {
            float f = (float)(cn.tri < 500u ? cn.tri : 500u) / 500.0f;
            out[at] = make_float4(f * 1.0f, f * 1.0f, f * 0.0f, f * 0.0f);
        } else if (fa.mode == RT_MODE_BOX_TESTS) {
            float f = (float)(cn.box < 1000u ? cn.box : 1000u) / 1000.0f;
            out[at] = make_float4(f * 0.0f, f * 1.0f, f * 1.0f, f * 1.0f);
        } else {
            uint32_t m = reflections < fa.bounces ? reflections : fa.bounces;
            float f = (float)m / (float)fa.bounces;
            out[at] = make_float4(f, f, f, f);
        }
    }
    if (COUNT && g_counters) flush_counters(cn, paths, g_counters);
}

// ---- wavefront: generate ---------------------------------------------------------------------------
__global__ __launch_bounds__(RT_BLOCK) void k_generate(DCamera cam, FrameArgs fa, DQueue q, uint32_t* counts,
                                                       unsigned long long* g_counters) {
    int gx, gy;
    tile_pixel(blockIdx.x, fa.tiles_x, threadIdx.x, gx, gy);
    size_t slot = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot < RT_WF_BOUNCES + 2) counts[slot] = slot == 0 ? (uint32_t)fa.num_tiles * RT_BLOCK : 0u;
    bool active = gx < cam.width && gy < cam.height;
    rt_vec3 o = rt_v3(0, 0, 0), d = rt_v3(0, 0, 1.0f), inv = rt_v3(0, 0, 0);
    uint32_t pix = 0xffffffffu, seed = 0;
    if (active) {
        primary_ray(cam, gx, gy, o, d, inv);
        seed = pixel_seed(cam, gx, gy, fa.frame);
        pix = (uint32_t)gy * (uint32_t)cam.width + (uint32_t)gx;
    }
    q.o[slot] = make_float4(o.x, o.y, o.z, rt_u2f(pix));
    q.d[slot] = make_float4(d.x, d.y, d.z, rt_u2f(seed));
    q.inv[slot] = make_float4(inv.x, inv.y, inv.z, 0.0f);
    q.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    if (g_counters) {  // one path per pixel on this pipeline
        Counters none = {0, 0, 0, 0, 0};
        flush_counters(none, active ? 1ull : 0ull, g_counters);
    }
}

// ---- wavefront: extend -----------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_extend(DScene sc, DQueue q, const uint32_t* n_ptr, float2* hits,
                                                     unsigned long long* g_counters, uint2* spill, size_t spill_stride) {
    __shared__ uint2 s_stack[RT_LDS_STACK][RT_BLOCK];
    uint32_t n = *n_ptr;
    size_t slot = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if ((size_t)blockIdx.x * RT_BLOCK >= n) return;  // whole block beyond the queue
    Counters cn = {0, 0, 0, 0, 0};
    if (slot < n) {
        float4 o = q.o[slot];
        if (rt_f2u(o.w) != 0xffffffffu) {
            float4 d = q.d[slot], inv = q.inv[slot];
            Stack st = make_stack(s_stack, spill, spill_stride, slot);
            float t;
            int id;
            closest_hit<COUNT>(sc, rt_v3(o.x, o.y, o.z), rt_v3(d.x, d.y, d.z), rt_v3(inv.x, inv.y, inv.z), st, t, id, cn);
            hits[slot] = make_float2(t, rt_u2f((uint32_t)id));
        }
    }
    if (COUNT && g_counters) flush_counters(cn, 0, g_counters);
}

// ---- wavefront: shade + compact --------------------------------------------------------------------
template <bool FIX>
__global__ __launch_bounds__(RT_BLOCK) void k_shade(DScene sc, FrameArgs fa, uint32_t bounce_index, DQueue qin,
                                                    const uint32_t* n_in_ptr, const float2* hits, DQueue qout,
                                                    uint32_t* n_out_ptr, const float4* prev, float4* out) {
    uint32_t n = *n_in_ptr;
    if ((size_t)blockIdx.x * RT_BLOCK >= n) return;
    size_t slot = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    bool alive = false;
    PathState p;
    uint32_t pix = 0xffffffffu;
    if (slot < n) {
        float4 o = qin.o[slot];
        pix = rt_f2u(o.w);
        if (pix != 0xffffffffu) {
            float4 d = qin.d[slot], inv = qin.inv[slot], thr = qin.thr[slot];
            float2 h = hits[slot];
            p.o = rt_v3(o.x, o.y, o.z);
            p.d = rt_v3(d.x, d.y, d.z);
            p.inv = rt_v3(inv.x, inv.y, inv.z);
            p.thr = rt_v3(thr.x, thr.y, thr.z);
            p.seed = rt_f2u(d.w);
            rt_vec3 result = rt_v3(0.0f, 0.0f, 0.0f);
            if (h.x < RT_INF) {
                alive = bounce<FIX>(sc, p, h.x, (int)rt_f2u(h.y), bounce_index);
                if (bounce_index + 1u >= fa.bounces) alive = false;  // the bounce loop is exhausted
            } else {
                result = rt_mul(sample_sky(p.d), p.thr);
            }
            if (!alive) store_pixel(fa, prev, out, (size_t)pix, result.x, result.y, result.z, 2.0f);
        }
    }
    // compaction: ballot + prefix count inside the wave, one atomic per wave
    unsigned long long mask = __ballot(alive);
    uint32_t total = (uint32_t)__popcll(mask);
    if (total) {
        uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        uint32_t base = 0;
        if (prefix == 0 && alive) base = atomicAdd(n_out_ptr, total);
        int leader = __ffsll((long long)mask) - 1;
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (alive) {
            size_t dst = (size_t)base + prefix;
            qout.o[dst] = make_float4(p.o.x, p.o.y, p.o.z, rt_u2f(pix));
            qout.d[dst] = make_float4(p.d.x, p.d.y, p.d.z, rt_u2f(p.seed));
            qout.inv[dst] = make_float4(p.inv.x, p.inv.y, p.inv.z, 0.0f);
            qout.thr[dst] = make_float4(p.thr.x, p.thr.y, p.thr.z, 0.0f);
        }
    }
}

// ---- wavefront: tail -------------------------------------------------------------------------------
// Finishes every path of a (small) queue: each lane loops extend + shade from bounce `first_bounce`.
template <bool COUNT, bool FIX>
__global__ __launch_bounds__(RT_BLOCK) void k_tail(DScene sc, FrameArgs fa, uint32_t first_bounce, DQueue q,
                                                   const uint32_t* n_ptr, const float4* prev, float4* out,
                                                   unsigned long long* g_counters, uint2* spill, size_t spill_stride) {
    __shared__ uint2 s_stack[RT_LDS_STACK][RT_BLOCK];
    uint32_t n = *n_ptr;
    Counters cn = {0, 0, 0, 0, 0};
    size_t lane_slot = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    Stack st = make_stack(s_stack, spill, spill_stride, lane_slot);
    for (size_t i = lane_slot; i < n; i += (size_t)gridDim.x * RT_BLOCK) {
        float4 o = q.o[i], d = q.d[i], inv = q.inv[i], thr = q.thr[i];
        PathState p;
        p.o = rt_v3(o.x, o.y, o.z);
        p.d = rt_v3(d.x, d.y, d.z);
        p.inv = rt_v3(inv.x, inv.y, inv.z);
        p.thr = rt_v3(thr.x, thr.y, thr.z);
        p.seed = rt_f2u(d.w);
        uint32_t pix = rt_f2u(o.w);
        rt_vec3 result = rt_v3(0.0f, 0.0f, 0.0f);
        for (uint32_t b = first_bounce; b < fa.bounces; b++) {
            float t;
            int id;
            closest_hit<COUNT>(sc, p.o, p.d, p.inv, st, t, id, cn);
            if (t < RT_INF) {
                if (!bounce<FIX>(sc, p, t, id, b)) break;
            } else {
                result = rt_mul(sample_sky(p.d), p.thr);
                break;
            }
        }
        store_pixel(fa, prev, out, (size_t)pix, result.x, result.y, result.z, 2.0f);
    }
    if (COUNT && g_counters) flush_counters(cn, 0, g_counters);
}

// ---- parity hook: primary hits -----------------------------------------------------------------------
__global__ __launch_bounds__(RT_BLOCK) void k_primary_hits(DScene sc, DCamera cam, int tiles_x, int* ids, float* ts,
                                                           uint2* spill, size_t spill_stride) {
    __shared__ uint2 s_stack[RT_LDS_STACK][RT_BLOCK];
    int gx, gy;
    tile_pixel(blockIdx.x, tiles_x, threadIdx.x, gx, gy);
    if (gx >= cam.width || gy >= cam.height) return;
    Stack st = make_stack(s_stack, spill, spill_stride, (size_t)blockIdx.x * RT_BLOCK + threadIdx.x);
    rt_vec3 o, d, inv;
    primary_ray(cam, gx, gy, o, d, inv);
    Counters cn = {0, 0, 0, 0, 0};
    float t;
    int id;
    closest_hit<false>(sc, o, d, inv, st, t, id, cn);
    size_t at = (size_t)gy * (size_t)cam.width + (size_t)gx;
    ids[at] = id;
    ts[at] = t;
}

__global__ void k_selftest_math(const float* x, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    out[i] = rt_log(v);
    out[(size_t)n + i] = rt_cos(v);
    out[2 * (size_t)n + i] = rt_exp(v);
    out[3 * (size_t)n + i] = rt_sqrt(v);
    out[4 * (size_t)n + i] = 1.0f / v;
    out[5 * (size_t)n + i] = rt_pow(v, 0.35f);
}

#endif  // RT_FRAME_KERNELS_H
