// Persistent path kernel: the default pipeline of the render path on MI355X.
//
// One launch renders a BATCH of frames.  The grid is two 1024-thread workgroups per CU (32 waves, 8 per SIMD); every
// wave is an independent worker that keeps its 64 lanes busy with paths drawn from a pool of (8x8 pixel tile, frame)
// items, so that lanes whose path has ended (60 % of primary rays leave the scene at once) do not idle while
// their neighbours traverse.  The items are numbered tile-major, frame-minor: the frames of a launch share the camera,
// so the waves that run at one time walk the same part of the BVH and a scene far larger than the L2 stays in it.  Each wave alternates two phases, each of which runs one kind of code on as many
// lanes as possible:
//
//   service   shade the lanes whose traversal has finished (bounce + Russian roulette, or the sample store),
//             refill empty lanes with new primary rays (wave ballot + mbcnt prefix over the empty lanes, tiles
//             handed out by a workgroup-local LDS counter -- no global atomics at all), and run the root-box
//             test of every new ray;
//   traverse  ordered BVH traversal; every iteration runs ONE kind of step (inner-node visit or triangle test),
//             chosen by a weighted vote of the lanes, until too few lanes are still traversing.
//
// Bounce rays are traversed the reference's way -- slab tests with the primary ray's stale 1/dir, so the walk follows the
// primary ray's line -- but a child whose box the ray's TRUE line misses by more than the scene's pad is dropped: no
// triangle in it can report a hit, so nothing changes except a third of the node visits (see cull_dist below).
//
// No single unit binds the kernel.  Measured on dragon80k: the vector ALUs issue ~0.34 instructions per cycle and SIMD (of
// ~0.4-0.5 sustainable), the vector-memory address unit is occupied 76 % of the time, waves spend 32 % of their cycles
// issuing and 36 % waiting for an instruction's operands; before the cull, removing 7 % of the vector instructions, adding
// 16 no-ops per step, staging no records in LDS (+78 % memory requests) or halving the resident waves each moved the time
// by only 0-18 %.  A wave instruction costs the same whatever its EXEC mask, so what pays is fewer loop iterations (more
// lanes served per iteration, fewer nodes visited) and the hardware maximum of 8 waves per SIMD (64 VGPRs).
//
// Memory placement (LDS is 160 KiB per CU, 80 KiB per workgroup):
//   traversal stacks   8 (ref, dist) entries per lane, [entry][thread] so 8-byte accesses never conflict: 64 KiB;
//                      deeper entries (rare) spill to HBM, coalesced across the wave
//   top of the BVH     the first PT_CACHE inner-node records (breadth-first, ~8 levels) are staged in LDS once
//                      per workgroup: 16 KiB, stored quad by quad (all first 16-byte quads, then all second ones, ...)
//                      so that lanes reading different records spread over all 16-byte bank slots
//   everything else    registers: ray, throughput, RNG state, pixel, bounce index, traversal cursor
//
// A finished path stores (throughput.rgb, dir.y) into a staging array [frame][pixel]; k_accumulate evaluates the
// sky for it (densely, one lane per pixel, instead of on the few lanes of a wave that happen to finish together)
// and folds the samples into the framebuffer in frame order, which keeps the running mean of raytrace.glsl:296
// bit-exact although frames are rendered concurrently.
#ifndef RT_PATHS_H
#define RT_PATHS_H

#include "rt_device.h"

#ifndef PT_BLOCK
#define PT_BLOCK 1024          // threads per workgroup (16 waves)
#define PT_BLOCKS_PER_CU 2     // -> 32 waves per CU, 8 per SIMD, the hardware maximum (needs <= 64 VGPRs)
#define PT_STACK 8             // traversal-stack entries per lane kept in LDS
#define PT_CACHE 254           // inner-node records staged in LDS (64 B each); 2 x (stacks+cache+counter) <= 160 KiB
#endif
// log2 of the distance in bytes between two stack slots of a lane (PT_BLOCK entries of 8 bytes)
#define PT_SLOT_SHIFT (PT_BLOCK == 1024 ? 13 : PT_BLOCK == 512 ? 12 : PT_BLOCK == 256 ? 11 : -1)
static_assert(PT_SLOT_SHIFT > 0 && (1 << PT_SLOT_SHIFT) == PT_BLOCK * 8, "PT_BLOCK must be 256, 512 or 1024");
#define PT_MIN_TRAV 40         // leave the traversal phase when fewer lanes than this still traverse ...
#define PT_MIN_SERVICE 16       // ... and keep servicing only while at least this many lanes need it
#define PT_W_INNER 1           // step choice: inner-node visit when PT_W_INNER * (lanes at inner nodes) >=
#define PT_W_LEAF 3            //              PT_W_LEAF * (lanes at leaves), triangle test otherwise (a triangle step is the
                               //              cheaper one and frees its lanes for the next inner steps)
#ifndef PT_MAX_BATCH
#define PT_MAX_BATCH 16        // frames per launch.  The end of a launch drains slowly (a few long paths keep their waves
                               // alive, ~0.3 ms on dragon80k); consecutive launches of a call overlap on two streams to hide it
#endif
#define PT_STRIP_W 16          // tile columns per vertical strip of the tile numbering (the last strip may stick out of the image)
#define PT_STRIPE_TILES 2      // interleaved bands: a stripe is 2 tile rows = 16 pixel rows

enum { ST_NEED = 0, ST_NEW = 1, ST_TRAV = 2, ST_DONE = 3 };

// What a launch of the path kernel does with primary rays.  The reference shoots the same primary ray for a pixel in every
// frame (no jitter, raytrace.glsl:279-288), so while the camera stands still its closest hit is the same in every frame:
//   PM_TRACE    every path traces its primary ray (what the reference does; the default)
//   PM_PRIMARY  one pass over the image that traces only the primary rays and stores their closest hits
//   PM_REUSE    every path starts from the stored hit of its pixel and traces only its bounce rays (RT_FLAG_REUSE_PRIMARY)
//   PM_QUEUE    the primary rays of the launch have been traced tile by tile by k_primary (rt_packets.h): every path
//               starts from the hit that kernel queued for it; paths whose primary ray left the scene are already finished
//   PM_MULTI    PM_TRACE for rays_per_pixel > 1: the samples of a pixel share the primary ray and continue ONE random-number
//               stream (raytrace.glsl:289-293), so they are traced one after the other by the lane that took the pixel
enum { PM_TRACE = 0, PM_PRIMARY = 1, PM_REUSE = 2, PM_QUEUE = 3, PM_MULTI = 4 };

struct PathsArgs {
    uint32_t frame_base, frame_stride, num_frames;
    uint32_t bounces;
    uint32_t rpp;                   // u_RaysPerPixel; > 1 only in PM_MULTI: samples are staged [frame][sample][pixel]
    int tiles_x, tiles_per_frame;   // 8x8 pixel tiles (of the band being rendered); tiles_per_frame counts whole strips
    int strip_tiles;                // tiles in one vertical strip: PT_STRIP_W * tile rows of the band
    int tile_y0;                    // first tile row of the band (contiguous band) or of its first stripe (interleaved)
    int stripe_period;              // 0: contiguous band; else tile rows between two stripes of the band (PT_STRIPE_TILES * bands)
    uint32_t total_tiles;           // num_frames * tiles_per_frame
    uint32_t pixels;                // width * height
    int cache_recs;                 // records actually staged (<= PT_CACHE)
    int min_trav, min_service;      // phase-switch thresholds (lanes)
    int w_inner, w_leaf;            // an inner step runs when w_inner * (lanes at inner nodes) >= w_leaf * (lanes at leaves)
    uint32_t chunk;                 // consecutive grabs of a workgroup take `chunk` consecutive items (1: every grab its own round)
    float cull_slack, cull_slope;   // true-direction cull of the inner-node step: slack < 0 off; else a box is widened by the pad
                                    // (cull_slack + cull_slope * reach of the box from the ray origin) / 2
    float2* hits;                   // [pixels] closest hit of every pixel's primary ray, (t, id bits): written in PM_PRIMARY, read in PM_REUSE
    float4* hitq;                   // [total_tiles][64] PM_QUEUE: (t, id bits, lane bits, 0) of the k-th primary ray of the item that hit something
    const float4* rnodes;           // k_primary: the inner-node records with the launch's ray origin subtracted from every box corner (k_rebase)
    const float4* rtris;            // k_primary: the triangle records rebased to the launch's ray origin (k_rebase_tris)
    uint32_t* pk_ctr;               // k_primary: its chunk counters, 16 words apart, set at launch (see k_primary)
    uint32_t pk_chunk;              // k_primary: items per grab
    uint32_t finite_boxes;          // k_primary: every box of the tree is finite (the sign-specific slab tests may be used)
    uint32_t* hcount;               // [total_tiles] PM_QUEUE: how many did; 0xffffffff: the item's primary rays are still to be traced, lane by lane
    float4* samples;                // [num_frames][pixels]
    uint2* spill;                   // [entries][gridDim.x * PT_BLOCK] or nullptr
    size_t spill_stride;
    unsigned long long* counters;
};

// Pixel (gx, gy) of position l (0..63) of work item `item`, and the item's frame index f within the launch.
// The frames of a launch share the camera: tile t of every frame is numbered before tile t+1 of any, so the waves that run
// at one time walk the same part of the BVH.  Tiles are numbered strip by strip (PT_STRIP_W tile columns wide, row-major
// inside), so that the tiles the chip works on at one time form a compact block of the image, not a full-width band.
// tile_pixel: the second half, from the tile's number tq = item / num_frames.
__device__ __forceinline__ void tile_pixel(const PathsArgs& a, uint32_t tq, int l, int& gx, int& gy) {
    const int t = (int)tq;
    const int strip = t / a.strip_tiles, r = t - strip * a.strip_tiles;
    const int ty = r / PT_STRIP_W, tx = strip * PT_STRIP_W + r % PT_STRIP_W;
    // tile row of the band -> tile row of the image
    const int gty = a.stripe_period ? (ty / PT_STRIPE_TILES) * a.stripe_period + (ty % PT_STRIPE_TILES) + a.tile_y0 : ty + a.tile_y0;
    gx = tx * 8 + (l & 7);
    gy = gty * 8 + (l >> 3);
}
__device__ __forceinline__ void item_pixel(const PathsArgs& a, uint32_t item, int l, uint32_t& f, int& gx, int& gy) {
    const uint32_t tq = item / a.num_frames;
    f = item - tq * a.num_frames;
    tile_pixel(a, tq, l, gx, gy);
}

__device__ __forceinline__ void flush_counters6(const Counters& cn, unsigned long long paths, unsigned long long* g) {
    unsigned long long v[6] = {cn.box, cn.tri, cn.rays, cn.hits, cn.pops, paths};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        unsigned long long x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g[k], x);
    }
}

// Spill accesses are volatile so that the compiler keeps them apart from the LDS accesses of the same
// stack slot instead of merging both into one flat access through a generic pointer.
__device__ __forceinline__ void spill_store(uint2* p, uint2 v) {
    volatile uint32_t* q = (volatile uint32_t*)p;
    q[0] = v.x;
    q[1] = v.y;
}
__device__ __forceinline__ uint2 spill_load(const uint2* p) {
    const volatile uint32_t* q = (const volatile uint32_t*)p;
    uint2 v;
    v.x = q[0];
    v.y = q[1];
    return v;
}


// The tail of a traversal step, hand-scheduled for the common case that no lane touches the stack spill area.
// v62/v63 are named explicitly because the 8-byte stack entry must sit in a register pair whose halves the code also
// uses one by one.
//
// pt_pop_lds: the lanes of `need` pop until an entry survives the `dist >= best` cull (raytrace.glsl:151) or their
// stack is empty (cur = RT_REF_IDLE).  stack0 = LDS byte address of the lane's slot 0; slots are 8 KiB apart.
__device__ __forceinline__ void pt_pop_lds(unsigned long long need, uint32_t stack0, float best, int& sp, int& cur) {
    unsigned long long save, np;
    uint32_t addr;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_and_b64 %[np], %[need], exec\n\t"
        "s_cbranch_scc0 Lpop_done%=\n"
        "Lpop_loop%=:\n\t"
        "s_mov_b64 exec, %[np]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[sp]\n\t"
        "v_cndmask_b32_e32 %[cur], %[idle], %[cur], vcc\n\t"
        "s_and_b64 %[np], %[np], vcc\n\t"
        "s_cbranch_scc0 Lpop_done%=\n\t"
        "s_mov_b64 exec, %[np]\n\t"
        "v_add_u32_e32 %[sp], -1, %[sp]\n\t"
        "v_lshl_add_u32 %[addr], %[sp], %[shift], %[stack0]\n\t"
        "ds_read_b64 v[62:63], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_nge_f32_e32 vcc, v63, %[best]\n\t"
        "v_cndmask_b32_e32 %[cur], %[cur], v62, vcc\n\t"
        "s_andn2_b64 %[np], %[np], vcc\n\t"
        "s_cbranch_scc1 Lpop_loop%=\n"
        "Lpop_done%=:\n\t"
        "s_mov_b64 exec, %[save]"
        : [save] "=&s"(save), [np] "=&s"(np), [addr] "=&v"(addr), [sp] "+v"(sp), [cur] "+v"(cur)
        : [need] "s"(need), [stack0] "v"(stack0), [best] "v"(best), [idle] "v"(RT_REF_IDLE), [shift] "n"(PT_SLOT_SHIFT)
        : "vcc", "scc", "v62", "v63", "memory");
}

// pt_inner_tail_lds: after the two box tests of an inner node (distances d1, d2, child references r1, r2): push the
// farther child if it can still hold a closer hit, continue with the nearer one or pop (raytrace.glsl:167-191; equal
// distances visit child 2 first).
__device__ __forceinline__ void pt_inner_tail_lds(float d1, float d2, int r1, int r2, uint32_t stack0, float best, int& sp, int& cur) {
    unsigned long long save, np;
    uint32_t addr;
    int rn;
    float dn;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmp_lt_f32_e32 vcc, %[d1], %[d2]\n\t"
        "v_cndmask_b32_e32 v62, %[r1], %[r2], vcc\n\t"
        "v_cndmask_b32_e32 v63, %[d1], %[d2], vcc\n\t"
        "v_cndmask_b32_e32 %[rn], %[r2], %[r1], vcc\n\t"
        "v_cndmask_b32_e32 %[dn], %[d2], %[d1], vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, v63, %[best]\n\t"
        "s_and_b64 exec, %[save], vcc\n\t"
        "v_lshl_add_u32 %[addr], %[sp], %[shift], %[stack0]\n\t"
        "ds_write_b64 %[addr], v[62:63]\n\t"
        "v_add_u32_e32 %[sp], 1, %[sp]\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[dn], %[best]\n\t"
        "v_cndmask_b32_e32 %[cur], %[cur], %[rn], vcc\n\t"
        "s_andn2_b64 %[np], %[save], vcc\n\t"
        "s_cbranch_scc0 Ltail_done%=\n"
        "Ltail_loop%=:\n\t"
        "s_mov_b64 exec, %[np]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[sp]\n\t"
        "v_cndmask_b32_e32 %[cur], %[idle], %[cur], vcc\n\t"
        "s_and_b64 %[np], %[np], vcc\n\t"
        "s_cbranch_scc0 Ltail_done%=\n\t"
        "s_mov_b64 exec, %[np]\n\t"
        "v_add_u32_e32 %[sp], -1, %[sp]\n\t"
        "v_lshl_add_u32 %[addr], %[sp], %[shift], %[stack0]\n\t"
        "ds_read_b64 v[62:63], %[addr]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_nge_f32_e32 vcc, v63, %[best]\n\t"
        "v_cndmask_b32_e32 %[cur], %[cur], v62, vcc\n\t"
        "s_andn2_b64 %[np], %[np], vcc\n\t"
        "s_cbranch_scc1 Ltail_loop%=\n"
        "Ltail_done%=:\n\t"
        "s_mov_b64 exec, %[save]"
        : [save] "=&s"(save), [np] "=&s"(np), [addr] "=&v"(addr), [rn] "=&v"(rn), [dn] "=&v"(dn), [sp] "+v"(sp), [cur] "+v"(cur)
        : [d1] "v"(d1), [d2] "v"(d2), [r1] "v"(r1), [r2] "v"(r2), [stack0] "v"(stack0), [best] "v"(best), [idle] "v"(RT_REF_IDLE), [shift] "n"(PT_SLOT_SHIFT)
        : "vcc", "scc", "v62", "v63", "memory");
}

// d, or RT_INF when the ray o + t * dir (t >= 0; ix, iy, iz ~ 1 / dir) passes the box at a distance: with
// slack = 2 * pad * max |1 / dir_k|, `hi + slack < lo` implies that the three slab intervals of the box widened by pad on
// every side have no common point in t >= 0 (each widened interval reaches at most slack / 2 beyond the plain one).
// v_min / v_max drop a NaN operand (0 * inf: the origin on a plane the ray runs parallel to); an infinite slack (a zero
// direction component, or a primary ray) never culls.
__device__ __forceinline__ float cull_dist(float d, rt_vec3 o, float ix, float iy, float iz, float slack, float nx, float ny,
                                           float nz, float xx, float xy, float xz) {
    const float ax = (nx - o.x) * ix, ay = (ny - o.y) * iy, az = (nz - o.z) * iz;
    const float bx = (xx - o.x) * ix, by = (xy - o.y) * iy, bz = (xz - o.z) * iz;
    const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), 0.0f));
    const float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return hi + slack < lo ? RT_INF : d;
}

// An upper bound (L1 norm) of the distance from o to the farthest point of a box.
__device__ __forceinline__ float box_reach(rt_vec3 o, float nx, float ny, float nz, float xx, float xy, float xz) {
    return (__builtin_fmaxf(__builtin_fabsf(nx - o.x), __builtin_fabsf(xx - o.x)) + __builtin_fmaxf(__builtin_fabsf(ny - o.y), __builtin_fabsf(xy - o.y))) +
           __builtin_fmaxf(__builtin_fabsf(nz - o.z), __builtin_fabsf(xz - o.z));
}

template <bool COUNT, bool FIX, int PMODE>
__global__ __launch_bounds__(PT_BLOCK, PT_BLOCK * PT_BLOCKS_PER_CU / 256) void k_paths(DScene sc, DCamera cam, PathsArgs a) {
    __shared__ uint2 s_stack[PT_STACK][PT_BLOCK];
    __shared__ float4 s_nodes[PT_CACHE * 4];
    __shared__ uint32_t s_next;
    const int tid = threadIdx.x;

    // quad q of record r sits at s_nodes[q * PT_CACHE + r]: a lane's four reads differ by constant offsets, and lanes
    // that read the same quad of different records spread over all eight 16-byte bank slots
    for (int i = tid; i < a.cache_recs * 4; i += PT_BLOCK) {
        int r = i >> 2, q = i & 3;
        s_nodes[q * PT_CACHE + r] = sc.nodes[i];
    }
    if (tid == 0) s_next = 0;
    __syncthreads();

    typedef float f4 __attribute__((ext_vector_type(4)));
    // byte offset of the staged records inside this workgroup's LDS allocation
    const uint32_t nodes_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&s_nodes[0];
    uint2* const spill = a.spill ? a.spill + ((size_t)blockIdx.x * PT_BLOCK + tid) : nullptr;
    const size_t spill_stride = a.spill_stride;
    // LDS stack entries are addressed with 32-bit LDS offsets and a 24-bit multiply: through the generic pointer the
    // compiler computes every entry address with a 64-bit v_mad_u64_u32, which issues at a quarter of the rate
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) u32x2 lds_uint2;
    const uint32_t stack_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2*)&s_stack[0][tid];
#define PT_SLOT(i_) ((lds_uint2*)(uintptr_t)(stack_lds + __umul24((uint32_t)(i_), (uint32_t)(PT_BLOCK * sizeof(uint2)))))
#define PT_PUT(i_, ref_, dist_)                                              \
    do {                                                                     \
        uint2 v_ = make_uint2((uint32_t)(ref_), rt_f2u(dist_));             \
        if ((i_) < PT_STACK)                                    \
            *PT_SLOT(i_) = u32x2{v_.x, v_.y};                                \
        else                                                                 \
            spill_store(spill + (size_t)((i_)-PT_STACK) * spill_stride, v_); \
    } while (0)
#define PT_GET(e_, i_)                                                       \
    do {                                                                     \
        if ((i_) < PT_STACK) {                                               \
            const u32x2 w_ = *PT_SLOT(i_);                                   \
            e_ = make_uint2(w_.x, w_.y);                                     \
        }                                                                    \
        else                                                                 \
            e_ = spill_load(spill + (size_t)((i_)-PT_STACK) * spill_stride); \
    } while (0)

    // Workgroups are dealt to the 8 XCDs round-robin by index.  Every round of gridDim.x work items is split into 8
    // consecutive runs, one per XCD, so that the (tile, frame) items an XCD works on at one time are neighbours and its
    // L2 holds one part of the scene instead of a copy of all of it.
    const uint32_t xcd_slot = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // wave-uniform cursor into the tile pool
    uint32_t tile_g = 0;
    int tile_off = 64, tile_cnt = 64;   // positions handed out / to hand out of the current item
    bool tile_traced = false;           // PM_QUEUE: the current item's primary rays have been traced by k_primary
    bool exhausted = false;

    // lane state
    int state = ST_NEED;
    PathState p;
    p.o = p.d = p.inv = p.thr = rt_v3(0.0f, 0.0f, 0.0f);
    p.seed = 0;
    uint32_t out_idx = 0, bnc = 0;
    uint32_t smp_left = 0;   // PM_MULTI: samples of the lane's pixel still to trace after the current one
    int cur = RT_REF_IDLE, sp = 0, best_id = -1;
    float best = RT_INF;
    Counters cn = {0, 0, 0, 0, 0};
    unsigned long long paths = 0;

#ifdef PT_DIAG
    unsigned long long dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// iteration counters (even k): only the first active lane counts; lane sums (odd k): every active lane adds 1
#define DIAG(k_, v_) dg[k_] += ((k_) & 1) ? 1ull : (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(true) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(true), 0u)) == 0u ? 1ull : 0ull)
#else
#define DIAG(k_, v_)
#endif

#ifdef PK_DEBUG
    uint32_t wd = 0;
#endif
    for (;;) {
        // ------------------------------------------------------------------ service phase
        for (;;) {
#ifdef PK_DEBUG
            if (++wd > 2000000u) {
                if ((tid & 63) == 0) printf("k_paths<%d>: wg %d stuck in service: tile_g %u off %d cnt %d exhausted %d state %d\n", PMODE, (int)blockIdx.x, tile_g, tile_off, tile_cnt, (int)exhausted, state);
                return;
            }
#endif
            DIAG(4, 1);
            if (state == ST_DONE || (state == ST_NEED && !exhausted)) { DIAG(5, 1); }
            if (PMODE == PM_PRIMARY) {
                if (state == ST_DONE) {  // num_frames is 1 in this mode: out_idx is the pixel index
                    a.hits[out_idx] = make_float2(best, rt_u2f((uint32_t)best_id));
                    state = ST_NEED;
                }
            } else if (state == ST_DONE) {
                // sample = sky(dir.y) * throughput, evaluated by k_accumulate; a path that ends without
                // reaching the sky (Russian roulette, bounce limit) contributes zero
                float4 sample = make_float4(p.thr.x, p.thr.y, p.thr.z, p.d.y);
                bool ended = true;
                if (best < RT_INF) {
                    if (COUNT) cn.hits++;
                    bool alive = bounce<FIX>(sc, p, best, best_id, bnc);
                    bnc++;
                    if (alive && bnc < a.bounces) {
                        ended = false;
                        state = ST_NEW;
                    }
                    sample = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                if (ended) {
                    a.samples[out_idx] = sample;
                    state = ST_NEED;
                    if (PMODE == PM_MULTI && smp_left != 0u) {
                        // the pixel's next sample: the same primary ray again (getColour takes the ray by value), the
                        // random-number stream goes on where the last path left it
                        smp_left--;
                        out_idx += a.pixels;
                        const uint32_t pix = out_idx % a.pixels;
                        const int gy = (int)(pix / (uint32_t)cam.width), gx = (int)(pix - (uint32_t)gy * (uint32_t)cam.width);
                        primary_ray(cam, gx, gy, p.o, p.d, p.inv);
                        p.thr = rt_v3(1.0f, 1.0f, 1.0f);
                        bnc = 0;
                        state = ST_NEW;
                        if (COUNT) paths++;
                    }
                }
            }
            const unsigned long long need = __builtin_amdgcn_ballot_w64(state == ST_NEED);
            if (need != 0ull && !exhausted) {
                // PM_QUEUE: an item holds the tile_cnt primary rays of its tile that hit something (none: the next item)
                while (tile_off >= tile_cnt) {
                    uint32_t k = 0;
                    if ((tid & 63) == 0) k = __hip_atomic_fetch_add(&s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    tile_g = a.chunk == 1u ? xcd_slot + k * gridDim.x : ((k / a.chunk) * gridDim.x + xcd_slot) * a.chunk + k % a.chunk;
                    tile_off = 0;
                    if (tile_g >= a.total_tiles) {
                        exhausted = true;
                        break;
                    }
                    if (PMODE == PM_QUEUE) {
                        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.hcount[tile_g]);
                        tile_traced = n != 0xffffffffu;
                        tile_cnt = tile_traced ? (int)n : 64;
                    }
                }
                if (!exhausted) {
                    int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    int avail = tile_cnt - tile_off;
                    if (state == ST_NEED && rank < avail) {
                        int l = tile_off + rank;
                        if (PMODE == PM_QUEUE && tile_traced) {
                            // the path starts where k_primary left it: at the closest hit of its primary ray
                            const float4 h = a.hitq[(size_t)tile_g * 64u + (uint32_t)l];
                            best = h.x;
                            best_id = (int)rt_f2u(h.y);
                            l = (int)rt_f2u(h.z);
                        }
                        uint32_t f;
                        int gx, gy;
                        item_pixel(a, tile_g, l, f, gx, gy);
                        if (gx < cam.width && gy < cam.height) {
                            primary_ray(cam, gx, gy, p.o, p.d, p.inv);
                            p.seed = pixel_seed(cam, gx, gy, a.frame_base + f * a.frame_stride);
                            p.thr = rt_v3(1.0f, 1.0f, 1.0f);
                            const uint32_t pix = (uint32_t)gy * (uint32_t)cam.width + (uint32_t)gx;
                            out_idx = (PMODE == PM_MULTI ? f * a.rpp : f) * a.pixels + pix;
                            if (PMODE == PM_MULTI) smp_left = a.rpp - 1u;
                            bnc = 0;
                            if (PMODE == PM_REUSE) {
                                // the primary ray's closest hit is known: the path goes straight to its first bounce (or the sky)
                                const float2 h = a.hits[pix];
                                best = h.x;
                                best_id = (int)rt_f2u(h.y);
                                cur = RT_REF_IDLE;
                                state = ST_DONE;
                                if (COUNT) paths++;
                            } else if (PMODE == PM_QUEUE && tile_traced) {
                                cur = RT_REF_IDLE;
                                state = ST_DONE;   // k_primary has counted the path and its primary ray
                            } else {
                                state = ST_NEW;
                                if (COUNT) paths++;
                            }
                        }
                    }
                    int taken = __builtin_popcountll(need);
                    tile_off += taken < avail ? taken : avail;
                }
            }
            if (state == ST_NEW) {
                if (COUNT) {
                    cn.rays++;
                    cn.box++;
                }
                float d0 = box_dist(p.o, p.inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2]);
                best = RT_INF;
                best_id = -1;
                if (d0 == RT_INF) {
                    state = ST_DONE;
                } else {
                    if (COUNT) cn.box++;
                    cur = sc.root_ref;
                    if (COUNT && cur < 0) cn.pops++;
                    sp = 0;
                    state = ST_TRAV;
                }
            }
            const unsigned long long pending = __builtin_amdgcn_ballot_w64(state == ST_DONE || (state == ST_NEED && !exhausted));
            if (pending == 0ull) break;
            if (__builtin_popcountll(pending) < a.min_service && __builtin_amdgcn_ballot_w64(cur != RT_REF_IDLE) != 0ull) break;
        }

        if (__builtin_amdgcn_ballot_w64(cur != RT_REF_IDLE) == 0ull) break;  // nothing traverses, nothing left to serve

        // ------------------------------------------------------------------ traversal phase
        // Every iteration runs ONE kind of step -- an inner-node visit or a triangle test -- chosen by
        // which of the two more lanes are waiting for; the other lanes keep their state and wait.
#define PT_POP()                                             \
    for (;;) {                                               \
        if (sp == 0) {                                       \
            cur = RT_REF_IDLE;                               \
            break;                                           \
        }                                                    \
        sp--;                                                \
        uint2 e_;                                            \
        PT_GET(e_, sp);                                      \
        if (!(rt_u2f(e_.y) >= best)) {                       \
            cur = (int)e_.x;                                 \
            if (COUNT && cur < 0) cn.pops++;                 \
            break;                                           \
        }                                                    \
    }
        for (;;) {
#ifdef PK_DEBUG
            if (++wd > 2000000u) {
                if ((tid & 63) == 0) printf("k_paths<%d>: wg %d stuck in traversal\n", PMODE, (int)blockIdx.x);
                return;
            }
#endif
            // idle lanes hold cur == RT_REF_IDLE (INT_MIN): inner <=> cur >= 0, leaf <=> cur in (INT_MIN, -1]
            const bool want_inner = cur >= 0;
            const bool want_leaf = (uint32_t)cur > 0x80000000u;
            const unsigned long long mi = __builtin_amdgcn_ballot_w64(want_inner), ml = __builtin_amdgcn_ballot_w64(want_leaf);
            const uint32_t ni = (uint32_t)__builtin_popcountll(mi), nl = (uint32_t)__builtin_popcountll(ml);
            if (ni + nl < (uint32_t)a.min_trav) {
                if (ni + nl == 0) break;
                // is there anything the service phase could do for the lanes that do not traverse?
                if (!exhausted || __builtin_amdgcn_ballot_w64(state == ST_DONE || (state == ST_TRAV && cur == RT_REF_IDLE)) != 0ull) break;
            }
            if (ni * (uint32_t)a.w_inner >= nl * (uint32_t)a.w_leaf) {
                if (want_leaf) { DIAG(3, 1); }
                if (state == ST_DONE || (state == ST_TRAV && cur == RT_REF_IDLE)) { DIAG(9, 1); }
                if (state == ST_NEED) { DIAG(11, 1); }
                if (want_inner) {
                    DIAG(0, 1);
                    DIAG(1, 1);
                    // Staged records come from LDS, the others from HBM, straight into the same four
                    // registers: one asm statement narrows EXEC to each group in turn.  (Written in C++
                    // the compiler fuses both sources into per-dword FLAT loads through a generic pointer,
                    // 13 memory instructions per visit instead of 4.)
                    f4 q0, q1, q2, q3;
                    {
                        const unsigned long long staged = __builtin_amdgcn_ballot_w64(cur < a.cache_recs);
                        const uint32_t cc = (uint32_t)cur;
                        const uint32_t base = nodes_lds + (cc << 4);
                        const uint32_t rec_off = cc << 6;  // byte offset from sc.nodes (the upload limits records to 4 GiB)
                        unsigned long long keep;
                        asm volatile(
                            "s_mov_b64 %4, exec\n\t"
                            "s_and_b64 exec, %4, %5\n\t"
                            "ds_read_b128 %0, %6\n\t"
                            "ds_read_b128 %1, %6 offset:%9\n\t"
                            "ds_read_b128 %2, %6 offset:%10\n\t"
                            "ds_read_b128 %3, %6 offset:%11\n\t"
                            "s_andn2_b64 exec, %4, %5\n\t"
                            "global_load_dwordx4 %0, %7, %8\n\t"
                            "global_load_dwordx4 %1, %7, %8 offset:16\n\t"
                            "global_load_dwordx4 %2, %7, %8 offset:32\n\t"
                            "global_load_dwordx4 %3, %7, %8 offset:48\n\t"
                            "s_mov_b64 exec, %4\n\t"
                            "s_waitcnt vmcnt(0) lgkmcnt(0)"
                            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&s"(keep)
                            : "s"(staged), "v"(base), "v"(rec_off), "s"(sc.nodes), "n"(PT_CACHE * 16), "n"(PT_CACHE * 32), "n"(PT_CACHE * 48)
                            : "memory");
                    }
                    if (COUNT) {
                        cn.pops++;
                        cn.box += 2;
                    }
                    int r1 = (int)rt_f2u(q3.x), r2 = (int)rt_f2u(q3.y);
                    float d1 = box_dist(p.o, p.inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                    float d2 = box_dist(p.o, p.inv, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                    // A bounce ray's slab tests run with the primary ray's stale 1/dir (raytrace.glsl:247-254 never refreshes
                    // it), so they lead the traversal along the primary ray's line, through subtrees the ray itself never
                    // comes near.  A child whose box, widened by the scene's pad, the ray's TRUE line misses cannot hold a
                    // hit: it is dropped here (distance = infinity: neither pushed nor entered), which leaves every result
                    // as it is and saves about a third of all node visits.  Primary rays are exempt: their slab test is
                    // the true one, and their origin may be far outside the scene, where the pad does not cover rounding.
                    if (!FIX && PMODE != PM_PRIMARY && a.cull_slack >= 0.0f) {
                        const float ix = __builtin_amdgcn_rcpf(p.d.x), iy = __builtin_amdgcn_rcpf(p.d.y), iz = __builtin_amdgcn_rcpf(p.d.z);
                        const float widest = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ix), __builtin_fabsf(iy)), __builtin_fabsf(iz));
                        const float slack = bnc != 0u ? a.cull_slack * widest : RT_INF;
                        float s1 = slack, s2 = slack;
                        if (a.cull_slope > 0.0f) {  // large scenes: the pad grows with the box's reach from the origin
                            const float per_reach = a.cull_slope * widest;
                            s1 = __builtin_fmaf(box_reach(p.o, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y), per_reach, slack);
                            s2 = __builtin_fmaf(box_reach(p.o, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w), per_reach, slack);
                        }
                        d1 = cull_dist(d1, p.o, ix, iy, iz, s1, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                        d2 = cull_dist(d2, p.o, ix, iy, iz, s2, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                    }
                    // Diagnostic builds only (tools/pad_sensitivity.sh): N independent vector or scalar no-ops per inner-node
                    // step, to measure what one more instruction in this loop costs.
#ifdef PT_PAD_VNOP
                    asm volatile(".rept " PT_PAD_VNOP "\n\tv_nop\n\t.endr" ::: "memory");
#endif
#ifdef PT_PAD_SNOP
                    asm volatile(".rept " PT_PAD_SNOP "\n\ts_nop 0\n\t.endr" ::: "memory");
#endif
                    // counting builds and waves with a lane at the spill area take the C++ tail
                    if (!COUNT && __builtin_amdgcn_ballot_w64(sp >= PT_STACK) == 0ull) {
                        pt_inner_tail_lds(d1, d2, r1, r2, stack_lds, best, sp, cur);
                    } else {
                        bool first_is_1 = d1 < d2;
                        int rn = first_is_1 ? r1 : r2, rf = first_is_1 ? r2 : r1;
                        float dn = first_is_1 ? d1 : d2, df = first_is_1 ? d2 : d1;
                        if (df < best) {
                            PT_PUT(sp, rf, df);
                            sp++;
                        }
                        if (dn < best) {
                            cur = rn;
                            if (COUNT && cur < 0) cn.pops++;
                        } else {
                            PT_POP();
                        }
                    }
                }
            } else {
                if (want_leaf) {
                    DIAG(6, 1);
                    DIAG(7, 1);
                    bool last = true;
                    if (cur != RT_REF_EMPTY) {
                        const int k = ~cur;
                        const float4* tr = (const float4*)((const char*)sc.tris + ((uint32_t)k << 6));  // 32-bit offset
                        float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
                        if (COUNT) cn.tri++;
                        float t = tri_dist_flat(p.o, p.d, rt_v3(t0.x, t0.y, t0.z), rt_v3(t1.x, t1.y, t1.z), rt_v3(t2.x, t2.y, t2.z));
                        if (t < best) {
                            best = t;
                            best_id = (int)rt_f2u(t1.w);
                        }
                        last = (rt_f2u(t0.w) & 1u) != 0u;
                        cur = ~(k + 1);
                    }
                    if (!COUNT && __builtin_amdgcn_ballot_w64(last && sp > PT_STACK) == 0ull) {
                        pt_pop_lds(__builtin_amdgcn_ballot_w64(last), stack_lds, best, sp, cur);
                    } else if (last) {
                        PT_POP();
                    }
                }
            }
        }
#undef PT_POP
        // a lane whose traversal ran out of nodes has its closest hit (or none): the loop only marks it idle
        if (state == ST_TRAV && cur == RT_REF_IDLE) state = ST_DONE;
    }
    if (COUNT && a.counters) flush_counters6(cn, paths, a.counters);
#ifdef PT_DIAG
    if (a.counters)
        for (int k = 0; k < 12; k++) {
            unsigned long long x = dg[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((tid & 63) == 0) atomicAdd(&a.counters[8 + k], x);
        }


#endif
#undef DIAG
#undef PT_PUT
#undef PT_GET
#undef PT_SLOT
}

// Folds the staged samples of one batch into the framebuffer, frame by frame in order:
// final = (acc*frameCount + colour) / (frameCount + 1)   (raytrace.glsl:296), or acc + colour in SUM mode.
// The pixels of a band, numbered 0 .. num_pixels-1 row by row: rows [row0, ...) for a contiguous band; for an
// interleaved band stripes of stripe_rows rows, period_rows apart, starting at row0.
struct BandPixels {
    uint32_t width, row0, stripe_rows, period_rows, num_pixels;
};

// rpp > 1: frame j's colour is the sum of its rpp staged samples in order, starting from BLACK = (0, 0, 0, 1), divided by rpp
// (raytrace.glsl:289-293).
__global__ __launch_bounds__(256) void k_accumulate(const float4* samples, uint32_t num_frames, uint32_t pixels, BandPixels bp,
                                                    uint32_t frame_base, uint32_t frame_stride, uint32_t sum_mode,
                                                    uint32_t read_prev, const float4* prev, float4* out, uint32_t rpp) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= bp.num_pixels) return;
    if (bp.period_rows) {
        const uint32_t lr = i / bp.width, x = i - lr * bp.width;
        i = ((lr / bp.stripe_rows) * bp.period_rows + lr % bp.stripe_rows + bp.row0) * bp.width + x;
    } else {
        i += bp.row0 * bp.width;
    }
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (read_prev) acc = prev[i];
    for (uint32_t j = 0; j < num_frames; j++) {
        float4 c;
        if (rpp == 1u) {
            float4 smp = samples[(size_t)j * pixels + i];
            rt_vec3 sky = rt_mul(sample_sky(rt_v3(0.0f, smp.w, 0.0f)), rt_v3(smp.x, smp.y, smp.z));
            c = make_float4(sky.x, sky.y, sky.z, 2.0f);
        } else {
            c = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            for (uint32_t k = 0; k < rpp; k++) {
                float4 smp = samples[((size_t)j * rpp + k) * pixels + i];
                rt_vec3 sky = rt_mul(sample_sky(rt_v3(0.0f, smp.w, 0.0f)), rt_v3(smp.x, smp.y, smp.z));
                // getColour returns result = 0 + sky * throughput, or 0 for a path that never reached the sky
                c = make_float4(c.x + (0.0f + sky.x), c.y + (0.0f + sky.y), c.z + (0.0f + sky.z), c.w + 1.0f);
            }
            const float r = (float)rpp;
            c = make_float4(c.x / r, c.y / r, c.z / r, c.w / r);
        }
        uint32_t frame = frame_base + j * frame_stride;
        float fc = (float)frame, fc1 = (float)(frame + 1u);
        bool s = sum_mode != 0;
        acc = make_float4(accumulate(acc.x, c.x, fc, fc1, s), accumulate(acc.y, c.y, fc, fc1, s),
                          accumulate(acc.z, c.z, fc, fc1, s), accumulate(acc.w, c.w, fc, fc1, s));
    }
    out[i] = acc;
}

#endif  // RT_PATHS_H
