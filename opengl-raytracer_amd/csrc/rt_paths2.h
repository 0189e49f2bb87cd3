// Persistent path kernel, second generation: a wavefront path tracer INSIDE every wave.
//
// STATUS: experimental, selected with the environment variable RT_PATHS_GEN=2; bit-identical to the other
// pipelines (the whole parity suite passes with it).  Measured on dragon80k@1080p it executes 26 % fewer vector
// instructions than rt_paths.h (256 M vs 347 M per frame, 57 % vs 48 % mean lane occupancy) but needs 98 VGPRs,
// i.e. 4 waves per SIMD instead of 6, and adds a dependent queue round trip per ray: 0.69 ms per frame against
// 0.57 ms.  The first-generation kernel therefore stays the default.
//
// The first persistent kernel (rt_paths.h) shades, regenerates and root-tests on whatever lanes happen to have
// finished -- 23 of 64 on average -- and that sparse "service" code is a quarter of all vector instructions of a
// kernel that is bound by VALU issue.  Here the expensive stages always run on full waves, fed by two small
// WAVE-PRIVATE ring queues in HBM (L2-resident; no other wave ever touches them, so there are no atomics, fences
// or races):
//
//   generate   all 64 lanes build the primary rays of one 8x8 pixel tile, whatever their own traversal state is
//              (the lane's ray stays in its registers), test them against the root box, store the sky sample of
//              those that miss, and append the others to the wave's READY queue (ballot + mbcnt rank);
//   extend     the lanes traverse as in rt_paths.h (one kind of step per iteration, stacks and top of the BVH in
//              LDS); a lane that finishes waits for the next light service, which runs as soon as a few lanes wait;
//   light service   a finished lane stores its sample (miss) or parks (ray, throughput, RNG state, hit) in the wave's
//              HIT queue, then pops its next ray from the READY queue -- some twenty instructions;
//   shade      when 64 hits are parked, all 64 lanes bounce them (normal flip, hemisphere sample, Russian roulette,
//              root-box test of the new ray) and append the survivors to the READY queue.
//
// "compact" is the ballot/mbcnt rank that packs survivors into the queues.  Samples are (throughput, dir.y) records;
// k_accumulate evaluates the sky and folds frames in order (rt_paths.h).
#ifndef RT_PATHS2_H
#define RT_PATHS2_H

#include "rt_paths.h"

#ifndef P2_BLOCK
#define P2_BLOCK 1024          // threads per workgroup
#define P2_BLOCKS_PER_CU 1
#define P2_STACK 8             // traversal-stack entries per lane kept in LDS
#define P2_CACHE 1500          // inner-node records staged in LDS
#endif
#define P2_RQ 256              // READY queue capacity per wave (entries of 4 float4)
#define P2_HQ 256              // HIT queue capacity per wave (entries of 5 float4)
#define P2_LIGHT 8             // run the light service when this many lanes have finished

struct Paths2Args {
    uint32_t frame_base, frame_stride, num_frames;
    uint32_t bounces;
    int tiles_x, tiles_per_frame;
    uint32_t total_tiles;
    uint32_t pixels;
    int cache_recs;
    int light;                      // P2_LIGHT (tunable)
    float4* samples;                // [num_frames][pixels]
    float4* queues;                 // per wave: 4*P2_RQ + 5*P2_HQ float4
    uint2* spill;
    size_t spill_stride;
    unsigned long long* counters;
};

#define P2_QUEUE_FLOAT4_PER_WAVE (4 * P2_RQ + 5 * P2_HQ)

template <bool COUNT, bool FIX>
__global__ __launch_bounds__(P2_BLOCK, P2_BLOCK* P2_BLOCKS_PER_CU / 256) void k_paths2(DScene sc, DCamera cam, Paths2Args a) {
    __shared__ uint2 s_stack[P2_STACK][P2_BLOCK];
    __shared__ float4 s_nodes[P2_CACHE * 4];
    __shared__ uint32_t s_next;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;

    for (int i = tid; i < a.cache_recs * 4; i += P2_BLOCK) {
        int r = i >> 2, q = i & 3;
        s_nodes[(r << 2) + (q ^ ((r >> 2) & 3))] = sc.nodes[i];
    }
    if (tid == 0) s_next = 0;
    __syncthreads();

    const uint32_t nodes_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&s_nodes[0];
    uint2* const spill = a.spill ? a.spill + ((size_t)blockIdx.x * P2_BLOCK + tid) : nullptr;
    const size_t spill_stride = a.spill_stride;
#define P2_PUT(i_, ref_, dist_)                                              \
    do {                                                                     \
        uint2 v_ = make_uint2((uint32_t)(ref_), rt_f2u(dist_));             \
        if ((i_) < P2_STACK)                                                 \
            s_stack[(i_)][tid] = v_;                                         \
        else                                                                 \
            spill_store(spill + (size_t)((i_)-P2_STACK) * spill_stride, v_); \
    } while (0)
#define P2_GET(e_, i_)                                                       \
    do {                                                                     \
        if ((i_) < P2_STACK)                                                 \
            e_ = s_stack[(i_)][tid];                                         \
        else                                                                 \
            e_ = spill_load(spill + (size_t)((i_)-P2_STACK) * spill_stride); \
    } while (0)

    // the wave's private queues: field-major so that consecutive entries are consecutive addresses
    const size_t wave_id = ((size_t)blockIdx.x * P2_BLOCK + tid) >> 6;
    float4* const rq = a.queues + wave_id * P2_QUEUE_FLOAT4_PER_WAVE;   // 4 fields x P2_RQ
    float4* const hq = rq + 4 * P2_RQ;                                   // 5 fields x P2_HQ
    uint32_t rq_head = 0, rq_tail = 0, hq_head = 0, hq_tail = 0;         // monotonic, wave-uniform
    bool exhausted = false;

    // lane state: the ray being traversed
    PathState p;
    p.o = p.d = p.inv = p.thr = rt_v3(0.0f, 0.0f, 0.0f);
    p.seed = 0;
    uint32_t out_idx = 0, bnc = 0;
    int cur = RT_REF_IDLE, sp = 0, best_id = -1;
    float best = RT_INF;
    bool done = false;  // traversal finished, result not yet handed to the light service
    Counters cn = {0, 0, 0, 0, 0};
    unsigned long long paths = 0;

    // Root-box test of a new ray and hand-over: sky sample for a miss, READY queue for the rest.
    // Executed by lanes with `valid`; all lanes of the wave must call it (ballot inside).
#define P2_EMIT(valid_, q_, oidx_, bnc_)                                                                              \
    do {                                                                                                              \
        bool go_ = false;                                                                                             \
        if (valid_) {                                                                                                 \
            if (COUNT) {                                                                                              \
                cn.rays++;                                                                                            \
                cn.box++;                                                                                             \
            }                                                                                                         \
            float d0_ = box_dist((q_).o, (q_).inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2]); \
            if (d0_ == RT_INF) {                                                                                      \
                a.samples[oidx_] = make_float4((q_).thr.x, (q_).thr.y, (q_).thr.z, (q_).d.y);                         \
            } else {                                                                                                  \
                if (COUNT) cn.box++;                                                                                  \
                go_ = true;                                                                                           \
            }                                                                                                         \
        }                                                                                                             \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(go_);                                              \
        if (go_) {                                                                                                    \
            const uint32_t r_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u)); \
            const uint32_t e_ = (rq_tail + r_) & (P2_RQ - 1);                                                         \
            rq[e_] = make_float4((q_).o.x, (q_).o.y, (q_).o.z, rt_u2f(oidx_));                                        \
            rq[P2_RQ + e_] = make_float4((q_).d.x, (q_).d.y, (q_).d.z, rt_u2f((q_).seed));                            \
            rq[2 * P2_RQ + e_] = make_float4((q_).inv.x, (q_).inv.y, (q_).inv.z, rt_u2f(bnc_));                       \
            rq[3 * P2_RQ + e_] = make_float4((q_).thr.x, (q_).thr.y, (q_).thr.z, 0.0f);                               \
        }                                                                                                             \
        rq_tail += (uint32_t)__builtin_popcountll(m_);                                                                \
    } while (0)

    for (;;) {
        // ================================================================== light service
        // 1. finished lanes: sample for a miss, HIT queue for a hit
        {
            const bool hit = done && best < RT_INF;
            if (done && !hit) a.samples[out_idx] = make_float4(p.thr.x, p.thr.y, p.thr.z, p.d.y);
            const unsigned long long mh = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                if (COUNT) cn.hits++;
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
                const uint32_t e = (hq_tail + r) & (P2_HQ - 1);
                hq[e] = make_float4(p.o.x, p.o.y, p.o.z, rt_u2f(out_idx));
                hq[P2_HQ + e] = make_float4(p.d.x, p.d.y, p.d.z, rt_u2f(p.seed));
                hq[2 * P2_HQ + e] = make_float4(p.inv.x, p.inv.y, p.inv.z, rt_u2f(bnc));
                hq[3 * P2_HQ + e] = make_float4(p.thr.x, p.thr.y, p.thr.z, best);
                hq[4 * P2_HQ + e] = make_float4(rt_u2f((uint32_t)best_id), 0.0f, 0.0f, 0.0f);
            }
            hq_tail += (uint32_t)__builtin_popcountll(mh);
            done = false;
        }
        // 2. keep the READY queue able to feed every idle lane: shade parked hits when a full wave of them
        //    waits, generate a tile of primary rays otherwise
        for (;;) {
            const uint32_t idle = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur == RT_REF_IDLE));
            const uint32_t ready = rq_tail - rq_head, parked = hq_tail - hq_head;
            const bool starving = ready < idle;
            bool shade = parked >= 64u || (starving && exhausted && parked != 0u);
            if (ready + 64u > P2_RQ) shade = false;  // never overflow the READY queue
            if (shade) {
                // ------------------------------------------------------------ shade (dense)
                const uint32_t n = parked < 64u ? parked : 64u;
                const bool have = lane < n;
                PathState q;
                q.o = q.d = q.inv = q.thr = rt_v3(0.0f, 0.0f, 0.0f);
                q.seed = 0;
                uint32_t q_out = 0, q_bnc = 0;
                bool alive = false;
                if (have) {
                    const uint32_t e = (hq_head + lane) & (P2_HQ - 1);
                    const float4 h0 = hq[e], h1 = hq[P2_HQ + e], h2 = hq[2 * P2_HQ + e], h3 = hq[3 * P2_HQ + e],
                                 h4 = hq[4 * P2_HQ + e];
                    q.o = rt_v3(h0.x, h0.y, h0.z);
                    q.d = rt_v3(h1.x, h1.y, h1.z);
                    q.inv = rt_v3(h2.x, h2.y, h2.z);
                    q.thr = rt_v3(h3.x, h3.y, h3.z);
                    q.seed = rt_f2u(h1.w);
                    q_out = rt_f2u(h0.w);
                    q_bnc = rt_f2u(h2.w);
                    alive = bounce<FIX>(sc, q, h3.w, (int)rt_f2u(h4.x), q_bnc);
                    q_bnc++;
                    if (!(alive && q_bnc < a.bounces)) {
                        alive = false;
                        a.samples[q_out] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // ended without reaching the sky
                    }
                }
                hq_head += n;
                P2_EMIT(alive, q, q_out, q_bnc);
                continue;
            }
            if (starving && !exhausted) {
                // ------------------------------------------------------------ generate (dense)
                uint32_t k = 0;
                if (lane == 0) k = __hip_atomic_fetch_add(&s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                const uint32_t tile_g = blockIdx.x + k * gridDim.x;
                if (tile_g >= a.total_tiles) {
                    exhausted = true;
                    continue;
                }
                const uint32_t f = tile_g / (uint32_t)a.tiles_per_frame;
                const int t = (int)(tile_g - f * (uint32_t)a.tiles_per_frame);
                const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
                const int gx = tx * 8 + (int)(lane & 7u), gy = ty * 8 + (int)(lane >> 3);
                const bool valid = gx < cam.width && gy < cam.height;
                PathState q;
                q.o = q.d = q.inv = rt_v3(0.0f, 0.0f, 0.0f);
                q.thr = rt_v3(1.0f, 1.0f, 1.0f);
                q.seed = 0;
                uint32_t q_out = 0;
                if (valid) {
                    primary_ray(cam, gx, gy, q.o, q.d, q.inv);
                    q.seed = pixel_seed(cam, gx, gy, a.frame_base + f * a.frame_stride);
                    q_out = f * a.pixels + (uint32_t)gy * (uint32_t)cam.width + (uint32_t)gx;
                    if (COUNT) paths++;
                }
                P2_EMIT(valid, q, q_out, 0u);
                continue;
            }
            break;
        }
        // 3. idle lanes pop their next ray
        {
            const bool empty = cur == RT_REF_IDLE;
            const unsigned long long me = __builtin_amdgcn_ballot_w64(empty);
            const uint32_t ready = rq_tail - rq_head;
            const uint32_t want = (uint32_t)__builtin_popcountll(me);
            const uint32_t n = want < ready ? want : ready;
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me, 0u));
            if (empty && r < n) {
                const uint32_t e = (rq_head + r) & (P2_RQ - 1);
                const float4 r0 = rq[e], r1 = rq[P2_RQ + e], r2 = rq[2 * P2_RQ + e], r3 = rq[3 * P2_RQ + e];
                p.o = rt_v3(r0.x, r0.y, r0.z);
                p.d = rt_v3(r1.x, r1.y, r1.z);
                p.inv = rt_v3(r2.x, r2.y, r2.z);
                p.thr = rt_v3(r3.x, r3.y, r3.z);
                p.seed = rt_f2u(r1.w);
                out_idx = rt_f2u(r0.w);
                bnc = rt_f2u(r2.w);
                cur = sc.root_ref;
                if (COUNT && cur < 0) cn.pops++;
                sp = 0;
                best = RT_INF;
                best_id = -1;
            }
            rq_head += n;
        }
        if (__builtin_amdgcn_ballot_w64(cur != RT_REF_IDLE) == 0ull) break;  // no ray anywhere: queues are empty too

        // ================================================================== extend
#define P2_POP()                                             \
    for (;;) {                                               \
        if (sp == 0) {                                       \
            done = true;                                     \
            cur = RT_REF_IDLE;                               \
            break;                                           \
        }                                                    \
        sp--;                                                \
        uint2 e_;                                            \
        P2_GET(e_, sp);                                      \
        if (!(rt_u2f(e_.y) >= best)) {                       \
            cur = (int)e_.x;                                 \
            if (COUNT && cur < 0) cn.pops++;                 \
            break;                                           \
        }                                                    \
    }
        for (;;) {
            const bool want_inner = cur >= 0;
            const bool want_leaf = (uint32_t)cur > 0x80000000u;
            const unsigned long long mi = __builtin_amdgcn_ballot_w64(want_inner), ml = __builtin_amdgcn_ballot_w64(want_leaf);
            const uint32_t ni = (uint32_t)__builtin_popcountll(mi), nl = (uint32_t)__builtin_popcountll(ml);
            if (ni + nl == 0u) break;
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(done)) >= (uint32_t)a.light) break;
            if (ni >= nl) {
                if (want_inner) {
                    f4 q0, q1, q2, q3;
                    {
                        const unsigned long long staged = __builtin_amdgcn_ballot_w64(cur < a.cache_recs);
                        const uint32_t cc = (uint32_t)cur;
                        const uint32_t base = nodes_lds + (cc << 6), x = (cc >> 2) & 3u;
                        const float4* rec = sc.nodes + (size_t)cc * 4;
                        unsigned long long keep;
                        asm volatile(
                            "s_mov_b64 %4, exec\n\t"
                            "s_and_b64 exec, %4, %5\n\t"
                            "ds_read_b128 %0, %6\n\t"
                            "ds_read_b128 %1, %7\n\t"
                            "ds_read_b128 %2, %8\n\t"
                            "ds_read_b128 %3, %9\n\t"
                            "s_andn2_b64 exec, %4, %5\n\t"
                            "global_load_dwordx4 %0, %10, off\n\t"
                            "global_load_dwordx4 %1, %10, off offset:16\n\t"
                            "global_load_dwordx4 %2, %10, off offset:32\n\t"
                            "global_load_dwordx4 %3, %10, off offset:48\n\t"
                            "s_mov_b64 exec, %4\n\t"
                            "s_waitcnt vmcnt(0) lgkmcnt(0)"
                            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&s"(keep)
                            : "s"(staged), "v"(base + (x << 4)), "v"(base + ((1u ^ x) << 4)), "v"(base + ((2u ^ x) << 4)),
                              "v"(base + ((3u ^ x) << 4)), "v"(rec)
                            : "memory");
                    }
                    if (COUNT) {
                        cn.pops++;
                        cn.box += 2;
                    }
                    float d1 = box_dist(p.o, p.inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                    float d2 = box_dist(p.o, p.inv, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                    int r1 = (int)rt_f2u(q3.x), r2 = (int)rt_f2u(q3.y);
                    bool first_is_1 = d1 < d2;
                    int rn = first_is_1 ? r1 : r2, rf = first_is_1 ? r2 : r1;
                    float dn = first_is_1 ? d1 : d2, df = first_is_1 ? d2 : d1;
                    if (df < best) {
                        P2_PUT(sp, rf, df);
                        sp++;
                    }
                    if (dn < best) {
                        cur = rn;
                        if (COUNT && cur < 0) cn.pops++;
                    } else {
                        P2_POP();
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
                        float t = tri_dist(p.o, p.d, rt_v3(t0.x, t0.y, t0.z), rt_v3(t1.x, t1.y, t1.z), rt_v3(t2.x, t2.y, t2.z));
                        if (t < best) {
                            best = t;
                            best_id = (int)rt_f2u(t1.w);
                        }
                        last = (rt_f2u(t0.w) & 1u) != 0u;
                        cur = ~(k + 1);
                    }
                    if (last) {
                        P2_POP();
                    }
                }
            }
        }
#undef P2_POP
    }
    if (COUNT && a.counters) flush_counters6(cn, paths, a.counters);
#undef P2_EMIT
#undef P2_PUT
#undef P2_GET
}

#endif  // RT_PATHS2_H
