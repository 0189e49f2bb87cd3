// librt_hip.so -- the render path on MI355X (gfx950): the C ABI of include/rt_api.h and its host side.
//
// Device code lives in the headers:
//   rt_paths.h          the default pipeline: one persistent kernel renders a batch of frames (k_paths), k_accumulate
//                       folds the staged samples into the framebuffer in frame order
//   rt_frame_kernels.h  the per-frame pipelines (wavefront: generate / extend / shade+compact / tail; megakernel: one
//                       lane per pixel, also used for the debug render modes), normals, the primary-hit parity hook
//   rt_build.h          BVH build, rebuild and refit on the device
//   rt_device.h         the shared per-ray routines (slab test, triangle test, ordered traversal, bounce, sky)
// All pipelines produce identical pixels.  There is no CPU fallback: without a GPU rt_create fails.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "rt_api.h"
#include "rt_device.h"
#include "rt_paths.h"
#include "rt_packets.h"
#include "rt_frame_kernels.h"
#include "rt_build.h"

#define RT_BFS_PREFIX 1500    // inner-node records laid out breadth-first at the front (PT_CACHE of them are staged in LDS)
#define RT_MAX_DEPTH 126      // deepest tree accepted (the reference's own stack holds 128 entries)
#ifndef RT_LANES
#define RT_LANES 2            // internal streams that single-frame calls rotate over (2, 4 and 6 measure the same, 3 is slower)
#endif

// Scratch of rt_raytrace_frame_noalloc: staged samples of one frame followed by the stack spill area.  Every GPU the
// library runs on holds its own copy of the array, at its own address, until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: scratch_for_device() resolves it once per device and every
// use below goes through the resolved pointer.
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[RT_MAX_DEVICES] = {};

// Device address of this device's copy of `scratch` (device must be current).  nullptr and *err set on failure.
static void* scratch_for_device(int device, hipError_t* err) {
    *err = hipSuccess;
    if (device < 0 || device >= RT_MAX_DEVICES) {
        *err = hipErrorInvalidDevice;
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            *err = e != hipSuccess ? e : hipErrorInvalidSymbol;
            return nullptr;
        }
        g_scratch_ptr[device] = ptr;
    }
    return g_scratch_ptr[device];
}

struct EventPair {
    hipEvent_t a, b;
    int cls;
};

struct BuiltData {
    int nv = 0, nf = 0, nv_cap = 0, nf_cap = 0;
    float* d_verts = nullptr;
    uint32_t* d_faces = nullptr;
    float4* d_fcol = nullptr;
    BuildState s{};
    int cur = 0;           // which of the ping-pong order arrays holds the final order
    int out_cap = 0;       // faces the output buffers (d_nodes, d_tris, d_shade) were sized for; 0 = not from a build
    int ref_cap = 0;       // faces the reference-layout buffers were sized for
    bool ref_valid = false;
    bool parents_valid = false;  // nparent filled (first refit after a build)
};

// the path kernel's cull pad of a scene, see cull_pad_of
struct CullPad {
    float slack = -1.0f;   // < 0: never cull; else 2 * pad (reach_slope == 0) or 2 * pad_near
    float slope = 0.0f;    // 2 * reach_slope
};

struct rt_context {
    int device = 0;
    bool finite_boxes = false;   // every box of the current tree has finite corners with min <= max (known from the upload's checks; the builder's boxes are unions of at least one triangle each)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    bool has_scene = false;
    DScene sc{};
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    float4* d_shade = nullptr;
    int num_tris = 0;
    int max_depth = 0;
    // a scene built on the device (rt_build_scene_device): topology and vertices kept for rebuilds
    BuiltData bd;
    int build_gen = 2;  // RT_BUILD_GEN=1 selects the first-generation level loop (no small-subtree kernel)
    int build_small_max = 256, build_tiny = 8;  // tuning: RT_BUILD_SMALL_MAX = 128 | 256, RT_BUILD_TINY = 0 | 4 | 8
    hipEvent_t build_ev0 = nullptr, build_ev1 = nullptr;
    // its reference-layout copy, made on demand for rt_get_built_scene
    float* d_ref_nodes = nullptr;
    float4 *d_ref_v0 = nullptr, *d_ref_v1 = nullptr, *d_ref_v2 = nullptr, *d_ref_col = nullptr;
    uint32_t* d_ref_src = nullptr;
    int32_t* d_ref_links = nullptr;
    int built_nodes = 0;
    float last_build_ms = 0.0f;
    // builder workspace, kept between builds (a per-frame rebuild must not pay for fresh allocations)
    std::vector<void*> ws_ptr;
    std::vector<size_t> ws_bytes;

    // per-resolution buffers
    size_t cap_slots = 0;  // queue capacity (padded pixel count)
    DQueue q[2] = {};
    float2* d_hits = nullptr;
    uint32_t* d_counts = nullptr;
    uint2* d_spill = nullptr;
    size_t spill_entries = 0;
    float4* d_frame = nullptr;  // framebuffer for the host-pointer entry points
    size_t frame_pixels = 0;

    // persistent path kernel
    int num_cus = 0;
    int cache_recs = 0;           // inner-node records that are laid out breadth-first at the front
    float4* d_samples = nullptr;  // [frames of one launch][pixels] staging
    size_t samples_bytes = 0;
    // RT_FLAG_REUSE_PRIMARY: closest hits of the primary rays of the last (camera, image size, band, scene) they were
    // traced for; later launches with the same key start their paths from them
    float2* d_phits = nullptr;
    size_t phits_pixels = 0;
    bool phits_valid = false;
    DCamera phits_cam{};
    uint32_t phits_band = 0;
    hipEvent_t phits_ready = nullptr;  // recorded on the context's stream after the pass that fills d_phits
    // primary rays traced per tile by k_primary (rt_packets.h): the hits it queues for the path kernel, one share per
    // pipeline slot like the staged samples
    float4* d_rnodes = nullptr;   // rebased inner-node records (k_rebase), one copy of rnodes_cap records per pipeline slot;
    size_t rnodes_cap = 0;        // allocated by the first launch that traces packets, kept while the scene's records fit
    float4* d_rtris = nullptr;    // rebased triangle records (k_rebase_tris), likewise
    size_t rtris_cap = 0;
    size_t num_tri_recs = 0;      // triangle records of the current scene (leaf-ordered; a malformed tree's overlapping leaves repeat triangles)
    size_t num_recs = 0;          // inner-node records of the current scene
    float4* d_hitq = nullptr;
    uint32_t* d_hcount = nullptr;
    uint32_t* d_pkctr = nullptr;               // k_primary's chunk counters: PK_COUNTERS x 16 words per staging slot
    size_t hitq_items = 0;        // (tile, frame) items the two buffers hold, all slots together
    uint2* d_pspill = nullptr;    // stack spill area, one copy per pipeline slot
    size_t pspill_entries = 0;
    // Launches of the path kernel are pipelined: consecutive launches (the frames of a frame-by-frame loop, the batches
    // of a multi-frame call) rotate over RT_LANES internal streams, so that the next one fills the chip while the last
    // long paths of a launch drain; only the accumulate kernels,
    // which carry the frame-to-frame dependency, are ordered on the context's stream.
    hipEvent_t prim_done[RT_LANES] = {};   // recorded on lane_stream[k] after the primary pass that filled slot k's hit queues
    bool prim_pending = false;             // prim_done[prim_last] has been recorded by a pipelined launch
    uint32_t prim_last = 0;
    hipStream_t lane_stream[RT_LANES] = {};
    hipEvent_t lane_done[RT_LANES] = {};   // recorded on lane_stream[k] after its path kernel
    hipEvent_t slot_free[RT_LANES] = {};   // recorded on the context's stream after the accumulate that read slot k
    uint32_t lane_next = 0;
    // Frames traced ahead of a steady frame-by-frame loop (rt_raytrace_frame called with the same camera and parameters
    // and consecutive frame counts, as the reference's loop does while the camera stands still): once the pattern has
    // held for a few calls, a call traces a batch of frames in one launch and the following calls only fold their staged
    // samples in, while the batch after it is already in flight.  A launch over several frames keeps the lanes busy where
    // a single-frame launch spends half its time draining.  Results are the same; what is traced ahead and never asked
    // for (at most one batch and the rest of another when the pattern breaks) is wasted work, so the batches start small.
    struct Ahead {
        bool have_last = false;            // cam / params / last_frame describe the previous eligible call
        DCamera cam;
        rt_params params;
        uint32_t last_frame = 0, streak = 0;
        bool valid = false, next_valid = false;    // staged: frames [base, base + count) in `slot`; [next_base, ...) in next_slot
        uint32_t base = 0, count = 0, slot = 0, next_base = 0, next_count = 0, next_slot = 0;
        uint32_t unjoined = 0;             // bit k: a path launch may be in flight on internal stream k that nothing on the context's stream waits for
    } ahead;
    int tune_ahead = 8;                    // RT_TUNE_AHEAD: most frames per traced-ahead batch; < 2: never trace ahead

    // scheduling parameters of the path kernel; a tuning build (-DRT_TUNABLES) lets RT_TUNE_* environment variables
    // override them at rt_create, the shipped library runs with these defaults
    // Cull pad of the current scene (see cull_pad_of): the pad by which the path kernel widens a child box before it skips
    // the subtree of a bounce ray that passes it at a distance; slack < 0: this scene is traversed uncut.
    CullPad cull;
    float tune_cull_slack = -2.0f;  // RT_TUNE_CULL_SLACK: >= -1 overrides the scene's value with a constant one (diagnostics)
    int tune_cull_reach = -1;       // RT_TUNE_CULL_REACH: 0 / 1 forces the constant / the reach-dependent pad
    int tune_min_trav = 0, tune_min_service = 0;   // phase-switch thresholds; 0: by the kind of launch (paths_thresholds)
    int tune_w_inner = PT_W_INNER, tune_w_leaf = PT_W_LEAF;
    int tune_cache = -1;  // >= 0: stage at most this many records in LDS
    // workgroups per CU of a launch that renders a single frame: with one, the launches of two consecutive frames are
    // resident side by side for their whole duration and each fills the issue slots the other's drain leaves idle
    int tune_frame_blocks = 1;
    int tune_frame_chunk = 1;                  // PathsArgs::chunk of such a launch
    int tune_batch_blocks = PT_BLOCKS_PER_CU;  // workgroups per CU of a pipelined launch of several frames
    int tune_overlap = 1;                      // 0: the batches of a call run one after the other on the context's stream (A/B runs)
    int tune_stagger = 1;                      // 1: the primary pass of a pipelined launch waits for the previous launch's (launch_paths_batch)
    int tune_queue_blocks = 1;                 // workgroups per CU of a pipelined path launch that starts from k_primary's hit queues
    int tune_last_blocks = PT_BLOCKS_PER_CU;   // ... of the last such launch of a call (no primary pass runs next to it)
    int tune_ahead_t4 = 6, tune_ahead_t8 = 40; // steady calls after which a frame loop is traced ahead 4 / tune_ahead frames at a time
    int tune_split = 0;                        // frames of a call over its launches: 0 full launches then the remainder, 1 evenly, 2 remainder first
    int tune_primary_chunk = 0;                // items a wave of k_primary takes per grab (0: by the launch's number of frames)
    int tune_primary_grid = 0;                 // > 0: k_primary's grid as is (tests of the counter arrangements for grids that are no multiple of 64 or 8)
    int tune_primary_blocks = PK_BLOCKS_PER_CU;  // workgroups per CU of k_primary's grid
    int tune_primary_joined = 5;                 // ... when the launch starts next to the previous launch's path kernel (RT_TUNE_PRIMARY_JOINED)
    int tune_signed_slabs = 1;                 // k_primary: sign-specific slab tests where a packet's rays allow them (0: A/B runs)
    int tune_packets = 1;                      // primary rays traced tile by tile: 1 by rule (use_packets), 0 never, 2 always (A/B runs)

    unsigned long long* d_counters = nullptr;
    unsigned char* scratch_ptr = nullptr;  // this device's copy of the static scratch array (rt_raytrace_frame_noalloc)

    bool profiling = false;
    std::vector<EventPair> pending;
    std::vector<EventPair> free_events;
    double ms[RT_NUM_KERNEL_CLASSES] = {};
    uint32_t launches[RT_NUM_KERNEL_CLASSES] = {};
};

static thread_local std::string g_create_error;

// Slack of the path kernel's cull test for a scene whose root box has diagonal D and whose triangles have edges
// e1 = b - a, e2 = c - a with |e1| * |e2| <= m1 and |e1|, |e2| <= m2.
//
// The kernel skips the subtree under a child box when a bounce ray's true line (origin o inside the root box, direction
// d, t >= 0) stays farther than `pad` from the box.  That never changes a result if no triangle inside the box can then
// report a hit, i.e. if the triangle test of raytrace.glsl:79-110, evaluated in fp32, accepts a ray only when the ray
// passes within `pad` of the triangle.  With u = 2^-24, E = RT_EPS and s = o - a (|s| <= D):
//   the test rejects |det| < E, so the computed barycentrics and distance differ from the exact solution of
//   o + t d = a + u e1 + v e2 by at most
//     |du| <= (7.5 |s| |e2| + 6.6 |e1||e2|) u / E + 2u,   |dv| likewise with e1,   |dt| <= (7.5 |s| + 6.5 T) |e1||e2| u / E + 2u T
//   (rounding of the two cross products, three dot products, the reciprocal and the products; T <= D + 2.04 m2 is the
//   largest distance to a point of the triangle), which moves the computed hit point by no more than
//     fp = (u / E) m1 (29 D + 26.5 m2) + 2u (D + 4.1 m2);
//   the test accepts barycentrics up to E outside [0, 1]: another 2 E (|e1| + |e2|) <= 4 E m2;
//   the stored edges are rounded differences (u m2), and the slab arithmetic of the cull test itself (a subtraction, an
//   approximate reciprocal, a product) is off by at most 4u D per plane.
// pad = fp + 4 E m2 + u m2 + 8u D.  (Until the bound was attacked numerically the floating-point part carried a safety
// factor of 4.  tools/cull_bound_probe.py aims 10^7 rays at the rims of grazing triangles -- |det| just above E, origins up
// to D away: the cases the derivation names as the worst -- runs the test in binary32 and measures in binary64 how far a
// ray reported as a hit really is from its triangle: at most 1/59 (dragon80k's D, m1, m2) and 1/52 (the teapot's) of
// fp + 4 E m2 + u m2.  The factor bought nothing the bound does not already give and cost 5 to 9 % of the throughput.)
// The first-order bounds need the relative error of det to be small, 6.5 u m1 / E << 1: scenes with m1 >= 2 (or a NaN
// anywhere) are not culled.
//
// |s| <= D is the worst case.  For the triangles inside ONE box |s| and T are at most the box's reach from the origin (the
// largest distance to any of its points), so the same bound with D replaced by the reach R gives a pad that grows with
// distance: pad(R) = reach_slope * R + pad_near.  The kernel evaluates it (with the L1 reach, which is never smaller) when
// reach_slope > 0; that costs 12 more vector instructions per step and pays where D is large against the triangles.
static CullPad cull_pad_of(double D, double m1, double m2, int reach_mode /* 0 never, 1 always, -1 by rule */) {
    const double u = 1.0 / 16777216.0, E = (double)RT_EPS;
    CullPad cp;
    if (!(D >= 0.0) || !(m1 < 2.0) || !(m2 < 1e18) || !(D < 1e18)) return cp;
    const double slope = ((u / E) * m1 * 29.0 + 2.0 * u) + 8.0 * u;
    const double near = ((u / E) * m1 * 26.5 * m2 + 2.0 * u * 4.1 * m2) + 4.0 * E * m2 + u * m2;
    const double pad = slope * D + near;   // the constant pad: every reach is at most D
    // by rule: where the constant pad exceeds a quarter of the longest triangle edge, i.e. the scene is large against its
    // triangles (measured: 9.27 M-triangle grid, pad 0.33 against edges <= 0.46: 1.05 -> 0.84 ms per frame with the
    // reach-dependent pad; dragon80k and teapot, pad 0.02 and 0.06 of the longest edge: 1 to 8 % slower with it)
    const bool reach = reach_mode < 0 ? 4.0 * pad > m2 : reach_mode > 0;
    cp.slack = (float)(2.0 * (reach ? near : pad) * 1.001);
    cp.slope = reach ? (float)(2.0 * slope * 1.001) : 0.0f;
    return cp;
}

static int fail(rt_context* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                                 \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail(ctx, RT_EDEVICE, "%s: %s", #call, hipGetErrorString(e_));       \
    } while (0)

template <class T>
static void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// A device allocation that lives for one call.
template <class T>
struct DevTmp {
    T* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
};

static void free_ref(rt_context* c) {
    dfree(c->d_ref_nodes);
    dfree(c->d_ref_v0);
    dfree(c->d_ref_v1);
    dfree(c->d_ref_v2);
    dfree(c->d_ref_col);
    dfree(c->d_ref_src);
    dfree(c->d_ref_links);
    c->bd.ref_cap = 0;
    c->bd.ref_valid = false;
}

static void free_scene(rt_context* c) {
    dfree(c->d_rnodes);
    c->rnodes_cap = 0;
    dfree(c->d_rtris);
    c->rtris_cap = 0;
    dfree(c->d_nodes);
    dfree(c->d_tris);
    dfree(c->d_shade);
    free_ref(c);
    c->bd.out_cap = 0;
    c->built_nodes = 0;
    c->has_scene = false;
    c->phits_valid = false;
}

static void free_buffers(rt_context* c) {
    for (int k = 0; k < 2; k++) {
        dfree(c->q[k].o);
        dfree(c->q[k].d);
        dfree(c->q[k].inv);
        dfree(c->q[k].thr);
    }
    dfree(c->d_hits);
    dfree(c->d_spill);
    c->cap_slots = 0;
    c->spill_entries = 0;
    dfree(c->d_samples);
    c->samples_bytes = 0;
    dfree(c->d_pspill);
    c->pspill_entries = 0;
    dfree(c->d_hitq);
    dfree(c->d_hcount);
    c->hitq_items = 0;
    dfree(c->d_phits);
    c->phits_pixels = 0;
    c->phits_valid = false;
}

// Brackets one launch with events when profiling is on.
struct LaunchTimer {
    rt_context* c;
    EventPair ep{};
    bool on;
    LaunchTimer(rt_context* ctx, int cls) : c(ctx), on(ctx->profiling) {
        if (!on) return;
        if (!c->free_events.empty()) {
            ep = c->free_events.back();
            c->free_events.pop_back();
        } else {
            (void)hipEventCreate(&ep.a);
            (void)hipEventCreate(&ep.b);
        }
        ep.cls = cls;
        (void)hipEventRecord(ep.a, c->stream);
    }
    ~LaunchTimer() {
        if (!on) return;
        (void)hipEventRecord(ep.b, c->stream);
        c->pending.push_back(ep);
    }
};

static int ensure_buffers(rt_context* c, int width, int height) {
    int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
    size_t slots = (size_t)tiles_x * tiles_y * RT_BLOCK;
    size_t min_slots = (size_t)RT_TAIL_BLOCKS * RT_BLOCK;
    if (slots < min_slots) slots = min_slots;
    size_t want_spill = c->max_depth > RT_LDS_STACK ? (size_t)(c->max_depth - RT_LDS_STACK + 1) : 0;
    if (slots <= c->cap_slots && want_spill <= c->spill_entries) return RT_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_buffers(c);
    HIPCHK(c, hipSetDevice(c->device));
    for (int k = 0; k < 2; k++) {
        HIPCHK(c, hipMalloc(&c->q[k].o, slots * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->q[k].d, slots * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->q[k].inv, slots * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->q[k].thr, slots * sizeof(float4)));
    }
    HIPCHK(c, hipMalloc(&c->d_hits, slots * sizeof(float2)));
    if (want_spill) HIPCHK(c, hipMalloc(&c->d_spill, want_spill * slots * sizeof(uint2)));
    c->cap_slots = slots;
    c->spill_entries = want_spill;
    return RT_OK;
}

static int check_render_args(rt_context* c, const rt_camera* cam, const rt_params* p) {
    if (!c) return RT_EINVAL;
    if (!cam || !p) return fail(c, RT_EINVAL, "camera and params must not be NULL");
    if (!c->has_scene) return fail(c, RT_ENOSCENE, "no scene uploaded");
    if (p->width <= 0 || p->height <= 0) return fail(c, RT_EINVAL, "width and height must be positive");
    if ((int64_t)p->width * p->height > (int64_t)16384 * 16384)
        return fail(c, RT_ELIMIT, "image larger than 16384x16384 pixels");
    if (p->rays_per_pixel == 0) return fail(c, RT_EINVAL, "rays_per_pixel must be at least 1");
    if (p->render_mode < RT_MODE_RENDER || p->render_mode > RT_MODE_REFLECTIONS)
        return fail(c, RT_EINVAL, "render_mode must be 1..4");
    if ((p->flags & RT_FLAG_REUSE_PRIMARY) && ((p->flags & (RT_FLAG_MEGAKERNEL | RT_FLAG_WAVEFRONT)) || p->render_mode != RT_MODE_RENDER ||
                                              p->rays_per_pixel != 1 || p->ray_bounces == 0))
        return fail(c, RT_EINVAL, "RT_FLAG_REUSE_PRIMARY is supported by the default pipeline only (render mode 1, rays_per_pixel 1)");
    if (p->band) {
        const uint32_t idx = p->band >> 16, cnt = p->band & 0x7fffu;
        if (cnt == 0 || idx >= cnt) return fail(c, RT_EINVAL, "band index %u out of range for %u bands", idx, cnt);
        if ((p->flags & (RT_FLAG_MEGAKERNEL | RT_FLAG_WAVEFRONT)) || p->render_mode != RT_MODE_RENDER || p->rays_per_pixel != 1 ||
            p->ray_bounces == 0)
            return fail(c, RT_EINVAL, "bands are supported by the default pipeline only (render mode 1, rays_per_pixel 1)");
    }
    return RT_OK;
}

static DCamera make_dcamera(const rt_camera* cam, int width, int height) {
    DCamera d;
    for (int k = 0; k < 3; k++) d.pos[k] = cam->pos[k];
    for (int k = 0; k < 9; k++) d.rot[k] = cam->rot[k];
    d.viewport_dist = cam->viewport_dist;
    // tan(u_FOV / 2.0f) * u_ViewportDist * 2.0f / u_ScreenWidth (raytrace.glsl:279); the tangent is
    // rounded once from double precision
    float tan_half = (float)std::tan((double)(cam->fov_rad / 2.0f));
    d.pixel_width = ((tan_half * cam->viewport_dist) * 2.0f) / (float)width;
    d.width_f = (float)width;
    d.half_w = (float)width / 2.0f;
    d.half_h = (float)height / 2.0f;
    d.width = width;
    d.height = height;
    return d;
}

// Buffers of the persistent path kernel: sample staging for one batch and the stack spill area.
// Frames rendered by one launch of the path kernel: PT_MAX_BATCH, fewer for large images so that the staged
// samples (16 bytes per pixel and frame, one slot of that many frames per internal stream) stay within 4 GiB -- except
// that a launch holds at least one frame, so images above 2^27 pixels stage RT_LANES x 16 bytes per pixel (8 GiB at the
// 16384 x 16384 limit).
static uint32_t paths_batch(size_t pixels) {
    const size_t fit = ((size_t)4 << 30) / RT_LANES / (pixels * sizeof(float4));
    return (uint32_t)(fit < 1 ? 1 : (fit > PT_MAX_BATCH ? PT_MAX_BATCH : fit));
}

// rays_per_pixel > 1 runs on the default pipeline as long as one frame's samples fit a launch's staging share.
static bool rpp_fits_paths(const rt_params* p) {
    return p->rays_per_pixel <= paths_batch((size_t)p->width * p->height);
}

// Frames traced ahead and then not asked for (the camera moved, the frame count jumped) are only forgotten: their
// launches run to the end on their internal streams, and whatever follows on such a stream -- the next launch into the
// same staging slot -- is ordered behind them by the stream itself.  The context's stream does not wait for them, so the
// first frame after a camera move is held up by the discarded work of ONE internal stream at most (the one its own
// launch goes to), not by both.  ahead.unjoined remembers that such a launch may still be running with nothing on the
// context's stream waiting for it: before anything that the caller may follow up with a free or a rewrite of what
// those launches read or write (scene upload, build, refit, staging reallocation, stream change, synchronise) the
// context's stream is made to wait (lanes_join).
static void lanes_join(rt_context* c) {
    for (int k = 0; k < RT_LANES; k++)
        if ((c->ahead.unjoined >> k & 1u) && c->lane_done[k]) (void)hipStreamWaitEvent(c->stream, c->lane_done[k], 0);
    c->ahead.unjoined = 0;
}
static void ahead_discard(rt_context* c) { c->ahead.valid = c->ahead.next_valid = false; }
static void ahead_reset(rt_context* c) {
    ahead_discard(c);
    lanes_join(c);
    c->ahead.have_last = false;
    c->ahead.streak = 0;
}

// Work items (8x8 tiles, whole strips of PT_STRIP_W tile columns) of one frame of a width x height image.
static size_t items_per_frame(int width, int height) {
    const size_t tx = ((size_t)(width + 7) / 8 + PT_STRIP_W - 1) / PT_STRIP_W * PT_STRIP_W;
    return tx * (size_t)((height + 7) / 8);
}

// `packets`: the launch traces its primary rays tile by tile and needs the hit queues too (as large again as the staged samples:
// a context that never does -- a scene of sub-pixel triangles -- never allocates them).
static int ensure_paths_buffers(rt_context* c, int width, int height, bool packets) {
    const size_t pixels = (size_t)width * height;
    size_t want_spill = c->max_depth > PT_STACK ? (size_t)(c->max_depth - PT_STACK + 1) : 0;
    // sized in bytes: a smaller image may render more frames per launch than the one the buffer was made for
    const uint32_t batch = paths_batch(pixels);
    const size_t want_bytes = (size_t)RT_LANES * batch * pixels * sizeof(float4);  // one slot of `batch` frames per lane
    const size_t want_items = packets ? (size_t)RT_LANES * batch * items_per_frame(width, height) : 0;
    if (want_bytes <= c->samples_bytes && want_spill <= c->pspill_entries && want_items <= c->hitq_items) return RT_OK;
    ahead_reset(c);  // launches traced ahead still write the staging area that is about to be freed
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (want_bytes > c->samples_bytes) {
        dfree(c->d_samples);
        c->samples_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_samples, want_bytes));
        c->samples_bytes = want_bytes;
    }
    if (want_items > c->hitq_items) {
        dfree(c->d_hitq);
        dfree(c->d_hcount);
        c->hitq_items = 0;
        HIPCHK(c, hipMalloc(&c->d_hitq, want_items * 64 * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_hcount, want_items * sizeof(uint32_t)));
        c->hitq_items = want_items;
    }
    if (want_spill > c->pspill_entries) {
        dfree(c->d_pspill);
        c->pspill_entries = 0;
        HIPCHK(c, hipMalloc(&c->d_pspill, RT_LANES * want_spill * (size_t)c->num_cus * PT_BLOCKS_PER_CU * PT_BLOCK * sizeof(uint2)));
        c->pspill_entries = want_spill;
    }
    return RT_OK;
}

// The rows of the band selected by params->band (the whole image for 0).
//   contiguous band i of n: tile rows [i*R/n, (i+1)*R/n), R = ceil(height / 8)
//   interleaved band i of n (RT_BAND_INTERLEAVED): the stripes of PT_STRIPE_TILES tile rows number i, i+n, i+2n, ...
struct BandLayout {
    int tile_y0 = 0, tile_rows = 0, stripe_period = 0;  // kernel view (tile rows)
    BandPixels px{};                                    // accumulate view (pixel rows)
    // pixel-row ranges [first, first + count) that make up the band, for transfers and clears
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
};

// want_ranges = false leaves `ranges` empty (and allocates nothing).
static BandLayout band_layout(const rt_params* p, bool want_ranges = true) {
    BandLayout b;
    const int H = p->height, W = p->width, R = (H + 7) / 8;
    const int idx = p->band ? (int)(p->band >> 16) : 0;
    const int cnt = p->band ? (int)(p->band & 0x7fffu) : 1;
    const bool inter = p->band && (p->band & RT_BAND_INTERLEAVED) && cnt > 1;
    b.px.width = (uint32_t)W;
    if (!inter) {
        const int t0 = (int)((int64_t)idx * R / cnt), t1 = (int)((int64_t)(idx + 1) * R / cnt);
        const int r0 = t0 * 8 < H ? t0 * 8 : H, r1 = t1 * 8 < H ? t1 * 8 : H;
        b.tile_y0 = t0;
        b.tile_rows = t1 - t0;
        b.px.row0 = (uint32_t)r0;
        b.px.num_pixels = (uint32_t)(r1 - r0) * (uint32_t)W;
        if (r1 > r0 && want_ranges) b.ranges.push_back({(uint32_t)r0, (uint32_t)(r1 - r0)});
        return b;
    }
    const int U = PT_STRIPE_TILES;
    b.tile_y0 = idx * U;
    b.stripe_period = U * cnt;
    b.px.row0 = (uint32_t)(idx * U * 8);
    b.px.stripe_rows = (uint32_t)(U * 8);
    b.px.period_rows = (uint32_t)(U * 8 * cnt);
    uint32_t rows = 0;
    for (int t = idx * U; t < R; t += U * cnt) {  // first tile row of each stripe
        const int te = t + U < R ? t + U : R;
        const int r0 = t * 8, r1 = te * 8 < H ? te * 8 : H;
        b.tile_rows += te - t;
        if (r1 > r0) {
            if (want_ranges) b.ranges.push_back({(uint32_t)r0, (uint32_t)(r1 - r0)});
            rows += (uint32_t)(r1 - r0);
        }
    }
    b.px.num_pixels = rows * (uint32_t)W;
    return b;
}

// Launch arguments of the path kernel for `nframes` frames frame_base, frame_base+stride, ... of the band `bl`;
// the caller adds the staging and spill pointers.
static void fill_paths_args(const rt_context* c, const rt_params* p, const BandLayout& bl, uint32_t frame_base, uint32_t stride,
                            uint32_t nframes, PathsArgs& a) {
    a.frame_base = frame_base;
    a.frame_stride = stride;
    a.num_frames = nframes;
    a.bounces = p->ray_bounces;
    a.rpp = p->rays_per_pixel;
    a.tiles_x = (p->width + 7) / 8;
    a.tile_y0 = bl.tile_y0;
    a.stripe_period = bl.stripe_period;
    a.strip_tiles = PT_STRIP_W * bl.tile_rows;
    a.tiles_per_frame = ((a.tiles_x + PT_STRIP_W - 1) / PT_STRIP_W) * a.strip_tiles;
    a.total_tiles = nframes * (uint32_t)a.tiles_per_frame;
    a.pixels = (uint32_t)((size_t)p->width * p->height);
    a.cache_recs = c->cache_recs < PT_CACHE ? c->cache_recs : PT_CACHE;
    a.min_trav = PT_MIN_TRAV;          // launch_k_paths sets the thresholds once the launch's mode is known
    a.min_service = PT_MIN_SERVICE;
    a.cull_slack = (p->flags & RT_FLAG_NO_CULL) ? -1.0f : (c->tune_cull_slack >= -1.0f ? c->tune_cull_slack : c->cull.slack);
    a.cull_slope = c->tune_cull_slack >= -1.0f ? 0.0f : c->cull.slope;
    a.w_inner = c->tune_w_inner;
    a.w_leaf = c->tune_w_leaf;
    a.chunk = nframes == 1 ? (uint32_t)c->tune_frame_chunk : 1u;
    if (c->tune_cache >= 0 && c->tune_cache < a.cache_recs) a.cache_recs = c->tune_cache;
    a.hits = nullptr;
    a.rnodes = nullptr;
    a.rtris = nullptr;
    a.pk_ctr = nullptr;
    a.pk_chunk = 1;
    a.finite_boxes = c->finite_boxes && c->tune_signed_slabs ? 1u : 0u;
    a.hitq = nullptr;
    a.hcount = nullptr;
    a.samples = nullptr;
    a.spill = nullptr;
    a.spill_stride = (size_t)c->num_cus * PT_BLOCKS_PER_CU * PT_BLOCK;
    a.counters = (p->flags & RT_FLAG_COUNTERS) ? c->d_counters : nullptr;
}

// overlapped: the launch shares the chip with the launch before or after it (pipelined calls); a single-frame launch then
// takes tune_frame_blocks workgroups per CU, any other launch fills the chip.
// When a wave leaves its traversal phase (fewer than min_trav lanes still traverse) and its service phase (fewer than
// min_service lanes wait).  Measured per kind of launch (tools/tune_queue*.sh; dragon80k / teapot / 3840x2160, ms per frame):
//   paths that start from k_primary's hit queues: every refill runs the bounce code, so a wave should serve rarely and
//     many lanes at once: 40/16 -> 10/6: 0.293 -> 0.268, 0.205 -> 0.189, 1.02 -> 0.93;
//   primary rays traced lane by lane, stale invDir: 40/16 -> 16/6: 0.301 -> 0.290, 0.231 -> 0.221, 1.078 -> 1.03
//     (9.27 M-triangle grid 0.99, unchanged);
//   RT_FLAG_FIX_INVDIR (long paths, most lanes carry bounce rays): 40/16 stays (16/6: 0.508 -> 0.527); behind k_primary's hit
//     queues 24/8 (bench.py --fix-invdir --steps 64: 7.26 -> 7.53 Grays/s).
static void paths_thresholds(const rt_context* c, bool fix, int pmode, PathsArgs& a) {
    a.min_trav = fix ? (pmode == PM_QUEUE ? 24 : PT_MIN_TRAV) : (pmode == PM_QUEUE || pmode == PM_REUSE ? 10 : 16);
    a.min_service = fix ? (pmode == PM_QUEUE ? 8 : PT_MIN_SERVICE) : 6;
    if (c->tune_min_trav > 0) a.min_trav = c->tune_min_trav;
    if (c->tune_min_service > 0) a.min_service = c->tune_min_service;
}

static void launch_k_paths(const rt_context* c, const DCamera& dc, const PathsArgs& args, bool count, bool fix, int pmode, bool overlapped,
                           hipStream_t st, bool followed = true) {
    PathsArgs a = args;
    paths_thresholds(c, fix, pmode, a);
    // (the spill area is sized for the full grid; a smaller grid uses a prefix of it)
    // Workgroups per CU.  A launch that has the chip to itself fills it (2 x 16 waves per CU).  Pipelined launches share it:
    // a single-frame launch takes one workgroup per CU, so that two consecutive frames are resident side by side; so does a
    // launch that starts from k_primary's hit queues AND is followed by another launch of the same call, which leaves half
    // of every CU's LDS and wave slots to that launch's primary pass -- that pass is dense arithmetic, this kernel waits on
    // memory most of the time, and next to each other they fill each other's gaps (dragon80k, bench.py --steps 20 / 128:
    // 11.8 -> 12.1 / 12.0 -> 12.4 Grays/s).  The last launch of a call has no primary pass next to it: with two workgroups
    // per CU a 16-frame queue launch alone takes 2.28 instead of 2.70 ms.
    const int per_cu = !overlapped                                ? PT_BLOCKS_PER_CU
                       : a.num_frames == 1                        ? c->tune_frame_blocks
                       : pmode == PM_QUEUE                        ? (followed ? c->tune_queue_blocks : c->tune_last_blocks)
                                                                  : c->tune_batch_blocks;
    dim3 grid(c->num_cus * per_cu), blk(PT_BLOCK);
#define PATHS(C_, F_, M_) hipLaunchKernelGGL((k_paths<C_, F_, M_>), grid, blk, 0, st, c->sc, dc, a)
    if (pmode == PM_PRIMARY) {  // no bounces in this pass: invDir handling does not matter
        if (count) PATHS(true, false, PM_PRIMARY); else PATHS(false, false, PM_PRIMARY);
    } else if (pmode == PM_REUSE) {
        if (count) { if (fix) PATHS(true, true, PM_REUSE); else PATHS(true, false, PM_REUSE); }
        else { if (fix) PATHS(false, true, PM_REUSE); else PATHS(false, false, PM_REUSE); }
    } else if (pmode == PM_MULTI) {
        if (count) { if (fix) PATHS(true, true, PM_MULTI); else PATHS(true, false, PM_MULTI); }
        else { if (fix) PATHS(false, true, PM_MULTI); else PATHS(false, false, PM_MULTI); }
    } else if (pmode == PM_QUEUE) {
        if (count) { if (fix) PATHS(true, true, PM_QUEUE); else PATHS(true, false, PM_QUEUE); }
        else { if (fix) PATHS(false, true, PM_QUEUE); else PATHS(false, false, PM_QUEUE); }
    } else {
        if (count) { if (fix) PATHS(true, true, PM_TRACE); else PATHS(true, false, PM_TRACE); }
        else { if (fix) PATHS(false, true, PM_TRACE); else PATHS(false, false, PM_TRACE); }
    }
#undef PATHS
}

// Does a launch with these parameters trace its primary rays tile by tile (k_primary) ahead of the path kernel?
// RT_FLAG_NO_CULL asks for the reference's literal lane-by-lane traversal; RT_FLAG_REUSE_PRIMARY has its own pass.
//
// A packet pays when the 64 rays of a tile visit much the same nodes, i.e. when triangles are large against a tile's
// footprint.  Estimated from what the host knows without looking at the mesh: triangle size ~ sqrt(surface of the root
// box / triangles), pixel footprint at the distance of the root box's centre.  Measured (tools/packets_ab.sh, ms per frame,
// lane by lane -> packets): teapot (ratio 51) 0.220 -> 0.185, dragon80k 1080p (7.6) 0.291 -> 0.23, 3840x2160 (15) 1.027 ->
// 0.80-0.83, with RT_FLAG_FIX_INVDIR 0.508 -> 0.491, 9.27 M-triangle grid (ratio 1: a tile's rays part ways a few levels
// down) 0.977 -> 0.968 -- no gain there, and the hit queues would double the staging memory.
static bool use_packets(const rt_context* c, const rt_params* p, const DCamera& dc) {
    if (c->tune_packets == 0 || (p->flags & (RT_FLAG_NO_CULL | RT_FLAG_REUSE_PRIMARY)) || p->rays_per_pixel != 1) return false;
    if (c->tune_packets >= 2 || (p->flags & RT_FLAG_PACKETS)) return true;
    const double ex = (double)c->sc.rmax[0] - c->sc.rmin[0], ey = (double)c->sc.rmax[1] - c->sc.rmin[1], ez = (double)c->sc.rmax[2] - c->sc.rmin[2];
    const double area = 2.0 * (ex * ey + ey * ez + ez * ex);
    double dist2 = 0.0;
    for (int k = 0; k < 3; k++) {
        const double m = 0.5 * ((double)c->sc.rmax[k] + c->sc.rmin[k]) - dc.pos[k];
        dist2 += m * m;
    }
    const double tri = std::sqrt(area / (double)(c->num_tris > 0 ? c->num_tris : 1));
    const double pixel = (double)dc.pixel_width / (double)dc.viewport_dist * std::sqrt(dist2);
    // calibrated on dragon80k at falling resolutions (tools/packets_threshold.sh; ratio, ms per frame lane by lane -> packets):
    // 7.6 0.293 -> 0.227, 5.1 0.147 -> 0.126, 3.8 0.096 -> 0.094, 3.2 0.070 -> 0.078, 2.5 0.046 -> 0.069; the teapot (ratio 13 and up)
    // gains or breaks even down to 480 x 270
    return tri >= 4.0 * pixel;   // NaN (degenerate box or camera): false
}

// Room for the rebased copies of the scene's inner-node and triangle records, one per pipeline slot.
static int ensure_rnodes(rt_context* c) {
    if (c->num_recs <= c->rnodes_cap && c->d_rnodes && c->num_tri_recs <= c->rtris_cap && c->d_rtris) return RT_OK;
    ahead_reset(c);   // launches in flight may still read the old copies
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < RT_LANES; k++) HIPCHK(c, hipStreamSynchronize(c->lane_stream[k]));
    if (!(c->num_recs <= c->rnodes_cap && c->d_rnodes)) {
        dfree(c->d_rnodes);
        c->rnodes_cap = 0;
        const size_t cap = c->num_recs + c->num_recs / 16 + 1;   // a rebuilt tree's record count wanders a little
        HIPCHK(c, hipMalloc(&c->d_rnodes, cap * RT_LANES * 4 * sizeof(float4)));
        c->rnodes_cap = cap;
    }
    if (!(c->num_tri_recs <= c->rtris_cap && c->d_rtris)) {
        dfree(c->d_rtris);
        c->rtris_cap = 0;
        const size_t cap = c->num_tri_recs + 1;
        HIPCHK(c, hipMalloc(&c->d_rtris, cap * RT_LANES * 4 * sizeof(float4)));
        c->rtris_cap = cap;
    }
    return RT_OK;
}

// Points the launch at pipeline slot `slot`'s share of the hit queues and traces the primary rays of its items on `st`;
// the path kernel that follows on the same stream runs in PM_QUEUE mode.
static void launch_k_primary(const rt_context* c, const DCamera& dc, PathsArgs& a, uint32_t slot, bool count, hipStream_t st, bool joins = false) {
    // the slot's copy of the inner-node records, box corners minus this launch's ray origin (the counting build reads the
    // records as they are)
    float4* const rn = c->d_rnodes + (size_t)slot * c->rnodes_cap * 4;
    a.rnodes = rn;
    float4* const rt = c->d_rtris + (size_t)slot * c->rtris_cap * 4;
    a.rtris = rt;
    if (!count && c->num_recs) {
        const uint32_t quads = (uint32_t)(c->num_recs * 4);
        hipLaunchKernelGGL(k_rebase, dim3((quads + 255u) / 256u), dim3(256), 0, st, (const float4*)c->d_nodes, rn, quads, dc.pos[0], dc.pos[1],
                           dc.pos[2]);
    }
    if (!count && c->num_tri_recs)
        hipLaunchKernelGGL(k_rebase_tris, dim3((unsigned)((c->num_tri_recs + 255u) / 256u)), dim3(256), 0, st, (const float4*)c->d_tris, rt,
                           (uint32_t)c->num_tri_recs, dc.pos[0], dc.pos[1], dc.pos[2]);
    const size_t share = c->hitq_items / RT_LANES;
    a.hitq = c->d_hitq + (size_t)slot * share * 64;
    a.hcount = c->d_hcount + (size_t)slot * share;
    // Alone on the chip the kernel fills it with eight workgroups per CU.  A launch that joins the previous launch's path kernel
    // (one 16-wave workgroup per CU) finds room for four per CU; more than five only queue behind those and start when the
    // items are nearly gone (bench.py, 256 steps: 16.1 Grays/s with 8 per CU, 16.6 with 5 or 6).
    dim3 grid(c->tune_primary_grid > 0 ? c->tune_primary_grid : c->num_cus * (joins ? c->tune_primary_joined : c->tune_primary_blocks)), blk(PK_BLOCK);
    a.pk_ctr = c->d_pkctr + (size_t)slot * (PK_COUNTERS * 16);
    // consecutive items are the frames of one tile (identical packets: the second one finds its records in the scalar cache),
    // so long chunks pay for launches of many frames; short launches need small ones for balance (dragon80k: 16-frame launches
    // 13.6 / 14.2 / 14.4 Grays/s with chunks of 2 / 3 / 4, single-frame launches 0.30 / 0.31 / 0.34 ms per frame with 1 / 2 / 4)
    a.pk_chunk = c->tune_primary_chunk > 0 ? (uint32_t)c->tune_primary_chunk : a.num_frames >= 16u ? 4u : a.num_frames >= 8u ? 2u : 1u;
    // every counter starts behind the chunks its waves take without asking
    (void)hipMemsetD32Async((hipDeviceptr_t)a.pk_ctr, (int)(grid.x / pk_counters(grid.x) * (PK_BLOCK / 64)), PK_COUNTERS * 16, st);
    if (count)
        hipLaunchKernelGGL((k_primary<true>), grid, blk, 0, st, c->sc, dc, a);
    else
        hipLaunchKernelGGL((k_primary<false>), grid, blk, 0, st, c->sc, dc, a);
}

static void launch_k_accumulate(const rt_context* c, const rt_params* p, const BandLayout& bl, const PathsArgs& a, const float4* prev,
                                float4* out) {
    const bool sum = (p->flags & RT_FLAG_SUM) != 0;
    const uint32_t read_prev = (sum || a.frame_base != 0) ? 1u : 0u;
    if (bl.px.num_pixels)
        hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((bl.px.num_pixels + 255) / 256)), dim3(256), 0, c->stream,
                           (const float4*)a.samples, a.num_frames, a.pixels, bl.px, a.frame_base, a.frame_stride, sum ? 1u : 0u,
                           read_prev, prev, out, a.rpp);
}

// RT_FLAG_REUSE_PRIMARY: makes sure c->d_phits holds the closest hits of the primary rays of (camera, size, band, current
// scene), tracing them in one pass of the path kernel on the context's stream if the last pass was for something else.
// Every path launch that reads them waits on c->phits_ready.
static int ensure_primary_hits(rt_context* c, const DCamera& dc, const rt_params* p, const BandLayout& bl) {
    const size_t pixels = (size_t)p->width * p->height;
    if (c->phits_valid && c->phits_pixels >= pixels && c->phits_band == p->band && std::memcmp(&c->phits_cam, &dc, sizeof dc) == 0)
        return RT_OK;
    if (pixels > c->phits_pixels) {
        HIPCHK(c, hipStreamSynchronize(c->stream));  // path kernels that read the old buffer are behind accumulates on this stream
        dfree(c->d_phits);
        c->phits_pixels = 0;
        HIPCHK(c, hipMalloc(&c->d_phits, pixels * sizeof(float2)));
        c->phits_pixels = pixels;
    }
    if (!c->phits_ready) HIPCHK(c, hipEventCreateWithFlags(&c->phits_ready, hipEventDisableTiming));
    c->phits_valid = false;
    PathsArgs a;
    fill_paths_args(c, p, bl, 0u, 1u, 1u, a);
    a.hits = c->d_phits;
    a.spill = c->d_pspill;  // first spill copy: every earlier path kernel has finished by the time this stream gets here
    // ... except those of internal streams whose accumulate has not been enqueued yet: there are none (a launch always
    // enqueues its accumulate), but a launch on an internal stream may still be RUNNING behind an accumulate that waits
    // for it, so wait for the internal streams explicitly
    for (int k = 0; k < RT_LANES; k++) HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane_done[k], 0));
    {
        LaunchTimer lt(c, RT_K_PATHS);
        launch_k_paths(c, dc, a, (p->flags & RT_FLAG_COUNTERS) != 0, false, PM_PRIMARY, false, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->phits_ready, c->stream));
    // the pass used the first spill copy: a later launch on an internal stream -- also one that does not read the hits and
    // so does not wait on phits_ready -- must not start on it before the pass has finished
    for (int k = 0; k < RT_LANES; k++) HIPCHK(c, hipEventRecord(c->slot_free[k], c->stream));
    std::memcpy(&c->phits_cam, &dc, sizeof dc);
    c->phits_band = p->band;
    c->phits_valid = true;
    return RT_OK;
}

// The primary pass of a pipelined launch, on the launch's internal stream `pst`.
// Primary passes run one after the other: this one starts when the previous launch's has finished, i.e. next to that
// launch's path kernel rather than next to its primary pass.  Two primary passes side by side only slow each other down,
// and the two path kernels after them then start together and drain together (bench.py --steps 20 on dragon80k:
// 9.9 -> 11.8 Grays/s).
static int launch_k_primary_pipelined(rt_context* c, const DCamera& dc, PathsArgs& a, uint32_t slot, hipStream_t pst, bool joins = false) {
    if (c->tune_stagger && c->prim_pending && c->prim_last != slot) HIPCHK(c, hipStreamWaitEvent(pst, c->prim_done[c->prim_last], 0));
    launch_k_primary(c, dc, a, slot, false, pst, joins);
    HIPCHK(c, hipEventRecord(c->prim_done[slot], pst));
    c->prim_pending = true;
    c->prim_last = slot;
    return RT_OK;
}

// Renders `nframes` (<= PT_MAX_BATCH) frames frame_base, frame_base+stride, ... concurrently and folds
// them, in order, into out (reading prev).  pipelined: the path kernel goes to one of the internal streams in turn
// (see rt_context::lane_stream) and overlaps with the previous launch's; the accumulate kernel, which carries the
// frame-to-frame dependency, goes to the context's stream.
static int launch_paths_batch(rt_context* c, const DCamera& dc, const rt_params* p, uint32_t frame_base,
                              uint32_t stride, uint32_t nframes, const float4* prev, float4* out, bool pipelined, bool followed = true,
                              bool joins = false) {
    const bool fix = (p->flags & RT_FLAG_FIX_INVDIR) != 0;
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    ahead_discard(c);  // this launch takes a staging slot of its own
    const bool packets = use_packets(c, p, dc);
    int rc = ensure_paths_buffers(c, p->width, p->height, packets);
    if (rc != RT_OK) return rc;
    // counters and per-kernel timing are read on the context's stream: such calls stay on it
    if (count || c->profiling || c->tune_overlap == 0) pipelined = false;
    const uint32_t slot = pipelined ? (c->lane_next++ % RT_LANES) : 0u;
    hipStream_t pst = pipelined ? c->lane_stream[slot] : c->stream;
    const BandLayout bl = band_layout(p);
    const bool reuse = (p->flags & RT_FLAG_REUSE_PRIMARY) != 0;
    if (reuse && (rc = ensure_primary_hits(c, dc, p, bl)) != RT_OK) return rc;
    PathsArgs a;
    fill_paths_args(c, p, bl, frame_base, stride, nframes, a);
    a.hits = reuse ? c->d_phits : nullptr;
    // every internal stream owns a fixed share of the staging area and its own spill copy, whatever the image size of
    // the launch: launches of different sizes that are in flight together never touch each other's samples
    a.samples = c->d_samples + (size_t)slot * (c->samples_bytes / RT_LANES / sizeof(float4));
    a.spill = c->d_pspill ? c->d_pspill + (size_t)slot * c->pspill_entries * a.spill_stride : nullptr;
    if (pipelined) {
        // the slot's previous reader (an accumulate on the context's stream) must have finished
        HIPCHK(c, hipStreamWaitEvent(pst, c->slot_free[slot], 0));
    } else {
        // slot 0 and the first spill copy may still be in use by an internal stream's kernel
        for (int k = 0; k < RT_LANES; k++) HIPCHK(c, hipStreamWaitEvent(pst, c->lane_done[k], 0));
        c->ahead.unjoined = 0;  // pst is the context's stream
    }
    if (reuse && pipelined) HIPCHK(c, hipStreamWaitEvent(pst, c->phits_ready, 0));
    if (packets && (rc = ensure_rnodes(c)) != RT_OK) return rc;
    if (packets && pipelined) {
        if ((rc = launch_k_primary_pipelined(c, dc, a, slot, pst, joins)) != RT_OK) return rc;
    } else if (packets) {
        LaunchTimer lt(c, RT_K_PRIMARY);
        launch_k_primary(c, dc, a, slot, count, pst);
    }
    {
        LaunchTimer lt(c, RT_K_PATHS);
        launch_k_paths(c, dc, a, count, fix, reuse ? PM_REUSE : packets ? PM_QUEUE : a.rpp > 1u ? PM_MULTI : PM_TRACE, pipelined, pst, followed);
    }
    HIPCHK(c, hipGetLastError());
    if (pipelined) {
        HIPCHK(c, hipEventRecord(c->lane_done[slot], pst));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane_done[slot], 0));
        c->ahead.unjoined &= ~(1u << slot);  // the wait also covers discarded launches that ran on this stream before
    }
    {
        LaunchTimer lt(c, RT_K_ACCUMULATE);
        launch_k_accumulate(c, p, bl, a, prev, out);
    }
    HIPCHK(c, hipGetLastError());
    // the staging slots this call used are free again once its accumulate has run
    if (pipelined) {
        HIPCHK(c, hipEventRecord(c->slot_free[slot], c->stream));
    } else {
        for (int k = 0; k < RT_LANES; k++) HIPCHK(c, hipEventRecord(c->slot_free[k], c->stream));
    }
    return RT_OK;
}

// Path kernel for frames [base, base + count) of a steady loop into staging slot `slot`, on that slot's internal stream.
static int launch_ahead(rt_context* c, const DCamera& dc, const rt_params* p, uint32_t base, uint32_t count, uint32_t slot) {
    const BandLayout bl = band_layout(p, false);
    PathsArgs a;
    fill_paths_args(c, p, bl, base, 1u, count, a);
    a.samples = c->d_samples + (size_t)slot * (c->samples_bytes / RT_LANES / sizeof(float4));
    a.spill = c->d_pspill ? c->d_pspill + (size_t)slot * c->pspill_entries * a.spill_stride : nullptr;
    hipStream_t pst = c->lane_stream[slot];
    HIPCHK(c, hipStreamWaitEvent(pst, c->slot_free[slot], 0));  // the slot's last reader has finished
    const bool packets = use_packets(c, p, dc);   // frame_call has made room for the rebased records
    int rc;
    if (packets && (rc = launch_k_primary_pipelined(c, dc, a, slot, pst)) != RT_OK) return rc;
    launch_k_paths(c, dc, a, false, (p->flags & RT_FLAG_FIX_INVDIR) != 0, packets ? PM_QUEUE : PM_TRACE, true, pst);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->lane_done[slot], pst));
    c->ahead.unjoined |= 1u << slot;
    return RT_OK;
}

// rt_raytrace_frame on the default pipeline.  Keeps track of whether the calls form a steady loop (see rt_context::Ahead)
// and, when they do, serves the frame from samples traced ahead or traces the next batch; otherwise one pipelined
// single-frame launch.
static int frame_call(rt_context* c, const DCamera& dc, const rt_params* p, uint32_t frame_count, const float4* prev, float4* out) {
    rt_context::Ahead& ah = c->ahead;
    if (use_packets(c, p, dc)) {   // before any launch is recorded as traced ahead: a reallocation drops what is in flight
        const int rc0 = ensure_rnodes(c);
        if (rc0 != RT_OK) return rc0;
    }
    const bool eligible = c->tune_ahead >= 2 && !c->profiling && !(p->flags & (RT_FLAG_COUNTERS | RT_FLAG_REUSE_PRIMARY)) &&
                          p->rays_per_pixel == 1;
    if (!eligible) {
        ahead_reset(c);
        return launch_paths_batch(c, dc, p, frame_count, 1u, 1u, prev, out, true);
    }
    const bool same = ah.have_last && std::memcmp(&ah.cam, &dc, sizeof dc) == 0 && std::memcmp(&ah.params, p, sizeof *p) == 0 &&
                      frame_count == ah.last_frame + 1u;
    ah.streak = same ? ah.streak + 1u : 0u;
    ah.cam = dc;
    ah.params = *p;
    ah.last_frame = frame_count;
    ah.have_last = true;
    if (!same) ahead_discard(c);

    const size_t pixels = (size_t)p->width * p->height;
    int rc;
    if (ah.valid && frame_count == ah.base + ah.count) {  // the current batch is used up: on to the one in flight, if any
        ah.valid = ah.next_valid;
        ah.base = ah.next_base;
        ah.count = ah.next_count;
        ah.slot = ah.next_slot;
        ah.next_valid = false;
    }
    if (ah.valid && !(frame_count >= ah.base && frame_count < ah.base + ah.count)) ahead_discard(c);
    // Batch size by how long the loop has been steady: what is traced ahead and never asked for is wasted, and a loop that
    // has run for 40 frames is likelier to run for 8 more than one that has run for 3.
    const uint32_t want = ah.streak < 2u ? 1u : ah.streak < (uint32_t)c->tune_ahead_t4 ? 2u : ah.streak < (uint32_t)c->tune_ahead_t8 ? 4u : (uint32_t)c->tune_ahead;
    const uint32_t cap = std::min<uint32_t>(paths_batch(pixels), std::min<uint32_t>(want, (uint32_t)c->tune_ahead));
    if (!ah.valid) {
        const uint32_t k = cap;
        if (k < 2u) return launch_paths_batch(c, dc, p, frame_count, 1u, 1u, prev, out, true);
        if ((rc = ensure_paths_buffers(c, p->width, p->height, use_packets(c, p, dc))) != RT_OK) return rc;
        ah.base = frame_count;
        ah.count = k;
        ah.slot = c->lane_next++ % RT_LANES;
        ah.valid = true;
        if ((rc = launch_ahead(c, dc, p, ah.base, ah.count, ah.slot)) != RT_OK) {
            ah.valid = false;
            return rc;
        }
    }
    const uint32_t j = frame_count - ah.base;
    // with the last frame of a batch the next batch goes out: a caller that issues its calls without waiting for them has
    // it in the queue long before the GPU gets there, and the two launches overlap on their streams
    if (j + 1u == ah.count && !ah.next_valid && cap >= 2u) {
        ah.next_base = ah.base + ah.count;
        ah.next_count = cap;
        ah.next_slot = c->lane_next++ % RT_LANES;
        if (ah.next_slot == ah.slot) ah.next_slot = (ah.slot + 1u) % RT_LANES;
        if ((rc = launch_ahead(c, dc, p, ah.next_base, ah.next_count, ah.next_slot)) != RT_OK) return rc;
        ah.next_valid = true;
    }
    // fold frame j of the current batch in, on the context's stream, after the batch's path kernel
    const BandLayout bl = band_layout(p);
    PathsArgs a;
    fill_paths_args(c, p, bl, frame_count, 1u, 1u, a);
    a.samples = c->d_samples + (size_t)ah.slot * (c->samples_bytes / RT_LANES / sizeof(float4)) + (size_t)j * pixels;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane_done[ah.slot], 0));
    c->ahead.unjoined &= ~(1u << ah.slot);  // everything launched on that internal stream so far now has a waiter on the context's stream
    {
        LaunchTimer lt(c, RT_K_ACCUMULATE);
        launch_k_accumulate(c, p, bl, a, prev, out);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->slot_free[ah.slot], c->stream));  // moved on with every frame folded in: the last one frees the slot
    return RT_OK;
}

extern "C" {

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_version(void) { return "rt_hip gfx950 " RT_BUILD_ID " " __DATE__; }

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(int device_id, rt_context** out_ctx) {
    if (!out_ctx) return fail(nullptr, RT_EINVAL, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, RT_EDEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(nullptr, RT_EINVAL, "device %d out of range [0,%d)", device_id, count);
    rt_context* c = new rt_context();
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreate(&c->own_stream)) != hipSuccess ||
        (e = hipMalloc(&c->d_counters, 32 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc(&c->d_counts, 64 * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(c->d_counters, 0, 32 * sizeof(unsigned long long))) != hipSuccess) {
        int rc = fail(nullptr, RT_EDEVICE, "context set-up failed: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    c->stream = c->own_stream;
    // this device's copy of the static scratch array, resolved once per device while that device is current
    c->scratch_ptr = (unsigned char*)scratch_for_device(device_id, &e);
    if (!c->scratch_ptr) {
        int rc = fail(nullptr, RT_EDEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device_id, hipGetErrorString(e));
        rt_destroy(c);
        return rc;
    }
    // The internal streams have the caller's default priority and must keep it.  Streams of one priority share a small
    // pool of hardware queues (4 per process by default): in a process that holds many streams -- PyTorch creates 32 with
    // its first torch.cuda.Stream -- an internal stream can land on the hardware queue of the other one or of the
    // caller's stream, and launches that are meant to overlap then run one after the other (frame-by-frame loop 0.54 ->
    // 0.67 ms per frame); callers that can should issue their own work on the context's stream (rt_get_stream) instead
    // of adding streams.  Lower-priority internal streams come from another queue pool and avoid that sharing, but they
    // invert the dependency order of this library: the context's stream (normal priority) holds a barrier packet that
    // waits for lane_done of a path kernel on a lower-priority queue, and every stream that in turn waits on the
    // context's stream -- a collective library's own stream does, through the event it records there -- holds another.
    // The queue arbiter serves a lower-priority queue only when no queue above it has a packet at its head, and a barrier
    // packet whose signal has not fired is such a packet: the kernel everybody waits for is never dispatched.  A process
    // with RCCL collectives hung that way on the GPU box with lowest-priority lanes, so the priority is not configurable.
    for (int k = 0; k < RT_LANES; k++) {
        if ((e = hipEventCreateWithFlags(&c->prim_done[k], hipEventDisableTiming)) != hipSuccess ||
            (e = hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&c->lane_done[k], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&c->slot_free[k], hipEventDisableTiming)) != hipSuccess) {
            int rc = fail(nullptr, RT_EDEVICE, "context set-up failed: %s", hipGetErrorString(e));
            rt_destroy(c);
            return rc;
        }
    }
    if ((e = hipMalloc(&c->d_pkctr, (size_t)RT_LANES * PK_COUNTERS * 16 * sizeof(uint32_t))) != hipSuccess) {
        int rc = fail(nullptr, RT_EDEVICE, "context set-up failed: %s", hipGetErrorString(e));
        rt_destroy(c);
        return rc;
    }
#ifdef RT_TUNABLES
    // Tuning builds only (librt_hip_tune.so, used by tools/ and by the tests that cross-check a non-default variant): the
    // scheduling parameters can be overridden from the environment, read once here.  The shipped library reads none.
    if (const char* e = getenv("RT_TUNE_W_INNER")) c->tune_w_inner = atoi(e);
    if (const char* e = getenv("RT_TUNE_W_LEAF")) c->tune_w_leaf = atoi(e);
    if (const char* e = getenv("RT_TUNE_MIN_TRAV")) c->tune_min_trav = atoi(e);
    if (const char* e = getenv("RT_TUNE_MIN_SERVICE")) c->tune_min_service = atoi(e);
    if (const char* e = getenv("RT_TUNE_CACHE")) c->tune_cache = atoi(e);
    if (const char* e = getenv("RT_TUNE_CULL_SLACK")) c->tune_cull_slack = (float)atof(e);  // -1: never cull
    if (const char* e = getenv("RT_TUNE_CULL_REACH")) c->tune_cull_reach = atoi(e);
    if (const char* e = getenv("RT_TUNE_AHEAD")) c->tune_ahead = atoi(e);
    if (const char* e = getenv("RT_TUNE_FRAME_BLOCKS")) c->tune_frame_blocks = atoi(e);
    if (const char* e = getenv("RT_TUNE_FRAME_CHUNK")) c->tune_frame_chunk = atoi(e);
    if (const char* e = getenv("RT_TUNE_PACKETS")) c->tune_packets = atoi(e);
    if (const char* e = getenv("RT_TUNE_BATCH_BLOCKS")) c->tune_batch_blocks = atoi(e);
    if (const char* e = getenv("RT_TUNE_OVERLAP")) c->tune_overlap = atoi(e);
    if (const char* e = getenv("RT_TUNE_STAGGER")) c->tune_stagger = atoi(e);
    if (const char* e = getenv("RT_TUNE_QUEUE_BLOCKS")) c->tune_queue_blocks = atoi(e);
    if (const char* e = getenv("RT_TUNE_AHEAD_T4")) c->tune_ahead_t4 = atoi(e);
    if (const char* e = getenv("RT_TUNE_AHEAD_T8")) c->tune_ahead_t8 = atoi(e);
    if (const char* e = getenv("RT_TUNE_SPLIT")) c->tune_split = atoi(e);
    if (const char* e = getenv("RT_TUNE_PRIMARY_CHUNK")) c->tune_primary_chunk = atoi(e);
    if (const char* e = getenv("RT_TUNE_PRIMARY_GRID")) c->tune_primary_grid = atoi(e);
    if (const char* e = getenv("RT_TUNE_PRIMARY_BLOCKS")) c->tune_primary_blocks = atoi(e);
    if (const char* e = getenv("RT_TUNE_PRIMARY_JOINED")) c->tune_primary_joined = atoi(e);
    if (const char* e = getenv("RT_TUNE_SIGNED_SLABS")) c->tune_signed_slabs = atoi(e);
    if (const char* e = getenv("RT_TUNE_LAST_BLOCKS")) c->tune_last_blocks = atoi(e);
    if (const char* e = getenv("RT_BUILD_GEN")) c->build_gen = atoi(e) == 1 ? 1 : 2;
    if (const char* e = getenv("RT_BUILD_SMALL_MAX")) c->build_small_max = atoi(e) == 128 ? 128 : 256;
    if (const char* e = getenv("RT_BUILD_TINY")) { int v = atoi(e); c->build_tiny = v == 0 ? 0 : (v == 4 ? 4 : 8); }
    // out-of-range values would stall the kernel's phase switching: a traversal threshold above the lanes that can
    // traverse while some wait for a refill would send the wave back and forth without progress
    auto clamp = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    c->tune_ahead = clamp(c->tune_ahead, 0, PT_MAX_BATCH);
    c->tune_min_trav = clamp(c->tune_min_trav, 0, 48);       // 0: by the kind of launch
    c->tune_min_service = clamp(c->tune_min_service, 0, 64);
    c->tune_w_inner = clamp(c->tune_w_inner, 1, 64);
    c->tune_w_leaf = clamp(c->tune_w_leaf, 1, 64);
    c->tune_frame_blocks = clamp(c->tune_frame_blocks, 1, PT_BLOCKS_PER_CU);
    c->tune_batch_blocks = clamp(c->tune_batch_blocks, 1, PT_BLOCKS_PER_CU);
    c->tune_queue_blocks = clamp(c->tune_queue_blocks, 1, PT_BLOCKS_PER_CU);
    c->tune_last_blocks = clamp(c->tune_last_blocks, 1, PT_BLOCKS_PER_CU);
    c->tune_primary_blocks = clamp(c->tune_primary_blocks, 1, PK_BLOCKS_PER_CU);
    c->tune_primary_joined = clamp(c->tune_primary_joined, 1, PK_BLOCKS_PER_CU);
    c->tune_primary_chunk = clamp(c->tune_primary_chunk, 0, 64);
    c->tune_primary_grid = clamp(c->tune_primary_grid, 0, 1 << 14);
    c->tune_frame_chunk = clamp(c->tune_frame_chunk, 1, 64);
#endif
    hipDeviceProp_t prop;
    c->num_cus = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
                     ? prop.multiProcessorCount : 256;
    *out_ctx = c;
    return RT_OK;
}

int rt_destroy(rt_context* c) {
    if (!c) return RT_EINVAL;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < RT_LANES; k++) {
        if (c->lane_stream[k]) {
            (void)hipStreamSynchronize(c->lane_stream[k]);
            (void)hipStreamDestroy(c->lane_stream[k]);
        }
        if (c->lane_done[k]) (void)hipEventDestroy(c->lane_done[k]);
        if (c->prim_done[k]) (void)hipEventDestroy(c->prim_done[k]);
        if (c->slot_free[k]) (void)hipEventDestroy(c->slot_free[k]);
    }
    free_scene(c);
    free_buffers(c);
    dfree(c->d_pkctr);
    if (c->phits_ready) (void)hipEventDestroy(c->phits_ready);
    for (void* p : c->ws_ptr)
        if (p) (void)hipFree(p);
    c->ws_ptr.clear();
    c->ws_bytes.clear();
    dfree(c->bd.d_verts);
    dfree(c->bd.d_faces);
    dfree(c->bd.d_fcol);
    if (c->build_ev0) (void)hipEventDestroy(c->build_ev0);
    if (c->build_ev1) (void)hipEventDestroy(c->build_ev1);
    dfree(c->d_frame);
    dfree(c->d_counters);
    dfree(c->d_counts);
    for (auto& ep : c->pending) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (auto& ep : c->free_events) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RT_OK;
}

int rt_set_stream(rt_context* c, void* hip_stream) {
    if (!c) return RT_EINVAL;
    ahead_reset(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RT_OK;
}

int rt_set_frames_ahead(rt_context* c, int32_t max_frames) {
    if (!c) return RT_EINVAL;
    if (max_frames < 0 || max_frames > PT_MAX_BATCH) return fail(c, RT_EINVAL, "max_frames must be in [0, %d]", PT_MAX_BATCH);
    ahead_reset(c);
    c->tune_ahead = max_frames;
    return RT_OK;
}

int rt_get_stream(rt_context* c, void** out_hip_stream) {
    if (!c) return RT_EINVAL;
    if (!out_hip_stream) return fail(c, RT_EINVAL, "out_hip_stream is NULL");
    *out_hip_stream = (void*)c->stream;
    return RT_OK;
}

int rt_synchronize(rt_context* c) {
    if (!c) return RT_EINVAL;
    lanes_join(c);  // frames traced ahead are in flight on the internal streams: "everything issued" includes them
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// Re-lays the reference's buffers out for the device (see rt_device.h) after validating that the node
// array is a tree whose indices stay in range -- a malformed scene must fail here, not fault on the GPU.
int rt_upload_scene(rt_context* c, const rt_scene* s) {
    if (!c) return RT_EINVAL;
    if (!s || !s->nodes || !s->triv0 || !s->triv1 || !s->triv2 || !s->colours)
        return fail(c, RT_EINVAL, "scene pointers must not be NULL");
    const int N = s->num_nodes, T = s->num_tris;
    if (N < 1 || T < 1) return fail(c, RT_EINVAL, "scene needs at least one node and one triangle");
    if (!s->links && (N > (1 << 24) || T > (1 << 24)))
        return fail(c, RT_ELIMIT, "float-encoded ids cannot address %d nodes / %d triangles: pass rt_scene.links", N, T);

    // Pass 1: validate the tree depth-first (first child first) and remember the visiting order.
    struct Visit {
        int node, depth, parent, slot;  // parent = position of the parent in `order` (-1: root)
        bool leaf;
        int a, b;
    };
    std::vector<Visit> order;
    order.reserve((size_t)N);
    std::vector<char> seen((size_t)N, 0);
    {
        struct Item {
            int node, depth, parent, slot;
        };
        std::vector<Item> todo;
        todo.push_back({0, 0, -1, 0});
        while (!todo.empty()) {
            Item it = todo.back();
            todo.pop_back();
            if (it.node < 0 || it.node >= N) return fail(c, RT_EINVAL, "node index %d out of range", it.node);
            if (seen[it.node]) return fail(c, RT_EINVAL, "node %d is referenced twice (not a tree)", it.node);
            seen[it.node] = 1;
            if (it.depth > RT_MAX_DEPTH) return fail(c, RT_ELIMIT, "tree deeper than %d levels", RT_MAX_DEPTH);
            const float* nd = s->nodes + (size_t)12 * it.node;
            Visit v;
            v.node = it.node;
            v.depth = it.depth;
            v.parent = it.parent;
            v.slot = it.slot;
            v.leaf = std::fabs(nd[8] - 1.0f) <= RT_EPS;  // raytrace.glsl:155
            v.a = s->links ? s->links[2 * (size_t)it.node] : (int)nd[9];  // raytrace.glsl:157-158,167-168
            v.b = s->links ? s->links[2 * (size_t)it.node + 1] : (int)nd[10];
            int pos = (int)order.size();
            if (v.leaf) {
                if (v.b >= v.a && (v.a < 0 || v.b >= T))
                    return fail(c, RT_EINVAL, "leaf %d: triangle range [%d,%d] out of range", it.node, v.a, v.b);
            } else {
                if (v.a < 0 || v.a >= N || v.b < 0 || v.b >= N)
                    return fail(c, RT_EINVAL, "node %d: child index out of range", it.node);
                todo.push_back({v.b, it.depth + 1, pos, 1});
                todo.push_back({v.a, it.depth + 1, pos, 0});
            }
            order.push_back(v);
        }
    }
    // Pass 2: record indices.  The first PT_CACHE inner nodes in BREADTH-first order get indices
    // 0..K-1 (the persistent kernel stages exactly that prefix in LDS); all others follow depth-first,
    // which keeps subtrees contiguous in memory.
    const int V = (int)order.size();
    std::vector<int> rec_of(V, -1), kids(2 * (size_t)V, -1);
    int num_inner = 0, max_depth = 0;
    for (int i = 0; i < V; i++) {
        if (order[i].parent >= 0) kids[2 * (size_t)order[i].parent + order[i].slot] = i;
        if (!order[i].leaf) num_inner++;
        if (order[i].depth > max_depth) max_depth = order[i].depth;
    }
    int next_rec = 0;
    {
        std::vector<int> level{0}, nxt;
        while (!level.empty() && next_rec < RT_BFS_PREFIX) {
            nxt.clear();
            for (int i : level) {
                if (order[i].leaf) continue;
                if (next_rec < RT_BFS_PREFIX) rec_of[i] = next_rec++;
                nxt.push_back(kids[2 * (size_t)i]);
                nxt.push_back(kids[2 * (size_t)i + 1]);
            }
            level.swap(nxt);
        }
    }
    const int cache_recs = next_rec;
    for (int i = 0; i < V; i++)
        if (!order[i].leaf && rec_of[i] < 0) rec_of[i] = next_rec++;
    // Pass 3: fill the records.
    // the path kernel addresses records and triangle records with 32-bit byte offsets: at most 2^26 of each
    if (num_inner >= (1 << 26) || T >= (1 << 26)) return fail(c, RT_ELIMIT, "more than 2^26 - 1 inner nodes or triangles");
    std::vector<float> recs((size_t)num_inner * 16, 0.0f);  // 16 floats per inner node
    std::vector<float> tris;                                 // 16 floats per leaf-ordered triangle
    tris.reserve((size_t)T * 16);
    int root_ref = 0;
    auto as_f = [](int v) { return rt_u2f((uint32_t)v); };
    // For the path kernel's cull test (cull_pad_of): the largest edge product and edge of any triangle, and whether
    // every box lies inside its parent's and every triangle inside its leaf's.  The reference's builder guarantees the
    // latter (bvh.cpp grows a box from its triangles); a caller's own tree that does not, or one with a NaN or infinite
    // vertex, is traversed uncut.
    double edge_prod_max = 0.0, edge_max = 0.0;
    bool nested = true;
    bool ordered = true;   // every box has min <= max along every axis (an empty leaf may come with any box)
    for (int i = 0; i < V; i++) {
        const Visit& v = order[i];
        const float* box = s->nodes + (size_t)12 * v.node;
        for (int j = 0; j < 3; j++) ordered = ordered && box[j] <= box[4 + j];
        if (v.parent >= 0) {
            const float* pb = s->nodes + (size_t)12 * order[v.parent].node;
            for (int j = 0; j < 3; j++) nested = nested && box[j] >= pb[j] && box[4 + j] <= pb[4 + j];
        }
        int ref;
        if (v.leaf) {
            if (v.b < v.a) {
                ref = RT_REF_EMPTY;
            } else {
                ref = ~(int)(tris.size() / 16);
                for (int k = v.a; k <= v.b; k++) {
                    const float *p0 = s->triv0 + 4 * (size_t)k, *p1 = s->triv1 + 4 * (size_t)k, *p2 = s->triv2 + 4 * (size_t)k;
                    float rec[16] = {p0[0], p0[1], p0[2], as_f(k == v.b ? 1 : 0),
                                     p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], as_f(k),
                                     p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2], 0.0f,
                                     0.0f, 0.0f, 0.0f, 0.0f};
                    tris.insert(tris.end(), rec, rec + 16);
                    for (int j = 0; j < 3; j++)
                        nested = nested && p0[j] >= box[j] && p0[j] <= box[4 + j] && p1[j] >= box[j] && p1[j] <= box[4 + j] &&
                                 p2[j] >= box[j] && p2[j] <= box[4 + j];
                    const double l1 = std::sqrt((double)rec[4] * rec[4] + (double)rec[5] * rec[5] + (double)rec[6] * rec[6]);
                    const double l2 = std::sqrt((double)rec[8] * rec[8] + (double)rec[9] * rec[9] + (double)rec[10] * rec[10]);
                    if (!std::isfinite(l1) || !std::isfinite(l2)) nested = false;
                    edge_prod_max = std::max(edge_prod_max, l1 * l2);
                    edge_max = std::max(edge_max, std::max(l1, l2));
                }
                // leaves of a malformed scene may overlap; the records they expand to obey the same limit
                if (tris.size() / 16 >= ((size_t)1 << 26)) return fail(c, RT_ELIMIT, "leaves expand to more than 2^26 - 1 triangle records");
            }
        } else {
            ref = rec_of[i];
        }
        if (v.parent < 0) {
            root_ref = ref;
        } else {
            float* pr = recs.data() + (size_t)16 * rec_of[v.parent];
            const float* mn = s->nodes + (size_t)12 * v.node;
            const float* mx = mn + 4;
            int o = v.slot == 0 ? 0 : 6;
            pr[o + 0] = mn[0]; pr[o + 1] = mn[1]; pr[o + 2] = mn[2];
            pr[o + 3] = mx[0]; pr[o + 4] = mx[1]; pr[o + 5] = mx[2];
            pr[12 + v.slot] = as_f(ref);
        }
    }

    HIPCHK(c, hipSetDevice(c->device));
    ahead_reset(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    size_t nrec = recs.size() / 16, ntri = tris.size() / 16;
    HIPCHK(c, hipMalloc(&c->d_nodes, (nrec ? nrec : 1) * 16 * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->d_tris, (ntri ? ntri : 1) * 16 * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->d_shade, (size_t)T * 2 * sizeof(float4)));
    if (nrec) HIPCHK(c, hipMemcpyAsync(c->d_nodes, recs.data(), nrec * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (ntri) HIPCHK(c, hipMemcpyAsync(c->d_tris, tris.data(), ntri * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // the normals kernel reads the caller's vertex arrays as the reference's does (normals.glsl:15-17)
    {
        DevTmp<float4> v0, v1, v2, col;
        const size_t vb = (size_t)T * sizeof(float4);
        HIPCHK(c, v0.alloc((size_t)T));
        HIPCHK(c, v1.alloc((size_t)T));
        HIPCHK(c, v2.alloc((size_t)T));
        HIPCHK(c, col.alloc((size_t)T));
        HIPCHK(c, hipMemcpyAsync(v0.p, s->triv0, vb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(v1.p, s->triv1, vb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(v2.p, s->triv2, vb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(col.p, s->colours, vb, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_normals, dim3((T + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, v0.p, v1.p, v2.p, col.p,
                           c->d_shade, T);
        hipError_t e = hipGetLastError();
        // the temporaries are released only after the stream has drained, also on failure
        hipError_t e2 = hipStreamSynchronize(c->stream);
        if (e != hipSuccess || e2 != hipSuccess) {
            free_scene(c);
            return fail(c, RT_EDEVICE, "scene upload: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        }
    }

    c->num_recs = nrec;
    c->num_tri_recs = ntri;
    c->sc.nodes = c->d_nodes;
    c->sc.tris = c->d_tris;
    c->sc.shade = c->d_shade;
    for (int k = 0; k < 3; k++) {
        c->sc.rmin[k] = s->nodes[k];
        c->sc.rmax[k] = s->nodes[4 + k];
    }
    c->sc.root_ref = root_ref;
    c->sc.num_tris = T;
    c->num_tris = T;
    c->max_depth = max_depth;
    c->cache_recs = cache_recs;
    {
        double diag2 = 0.0;
        for (int k = 0; k < 3; k++) diag2 += ((double)s->nodes[4 + k] - s->nodes[k]) * ((double)s->nodes[4 + k] - s->nodes[k]);
        c->cull = nested ? cull_pad_of(std::sqrt(diag2), edge_prod_max, edge_max, c->tune_cull_reach) : CullPad();
        c->finite_boxes = nested && ordered && std::isfinite(diag2);   // nested: every box inside its parent's; comparisons are false on a NaN
    }
    c->has_scene = true;
    // a deeper tree may need a larger spill area: force re-allocation on the next render
    if (c->max_depth > RT_LDS_STACK && (size_t)(c->max_depth - RT_LDS_STACK + 1) > c->spill_entries) free_buffers(c);
    return RT_OK;
}

// Cull pad of a device-built tree: its boxes are exact unions of what is below them, the triangle emit kernel has
// reduced the edge maxima (a NaN among them reads back as a NaN).
static CullPad built_cull_pad(float4 rlo, float4 rhi, const float edge_max[2], int reach_mode) {
    const double dx = (double)rhi.x - rlo.x, dy = (double)rhi.y - rlo.y, dz = (double)rhi.z - rlo.z;
    if (!std::isfinite(edge_max[0]) || !std::isfinite(edge_max[1])) return CullPad();
    // the kernel's lengths are fp32: a relative 1e-6 covers their rounding
    return cull_pad_of(std::sqrt(dx * dx + dy * dy + dz * dz), (double)edge_max[0] * 1.000001, (double)edge_max[1] * 1.000001, reach_mode);
}

// ---- GPU BVH build (rt_build.h) ---------------------------------------------------------------------------
// Builds from the topology kept in the context (c->bd) and the vertices in c->bd.d_verts.  All device memory comes from
// the cached workspace or from output buffers sized for the worst case, so that a rebuild allocates nothing.
static int build_core(rt_context* c) {
    ahead_reset(c);  // nothing traced ahead may still read the tree that is about to be rewritten
    BuiltData& bd = c->bd;
    const int nf = bd.nf, max_nodes = 2 * nf;
    const bool v1 = c->build_gen == 1;  // first-generation level loop all the way down (kept as the cross-check)
    const int small_max = v1 ? 0 : c->build_small_max;
    const size_t max_active = v1 ? (size_t)nf / 2 + 2 : (size_t)nf / (size_t)small_max + 1024;
    hipStream_t st = c->stream;
    BuildState& s = bd.s;
    size_t ws_next = 0;
    auto dalloc = [&](void** p, size_t bytes) -> hipError_t {  // slot ws_next of the cached workspace, grown on demand
        const size_t k = ws_next++;
        if (c->ws_ptr.size() <= k) {
            c->ws_ptr.resize(k + 1, nullptr);
            c->ws_bytes.resize(k + 1, 0);
        }
        if (c->ws_bytes[k] < bytes || !c->ws_ptr[k]) {
            if (c->ws_ptr[k]) (void)hipFree(c->ws_ptr[k]);
            c->ws_ptr[k] = nullptr;
            c->ws_bytes[k] = 0;
            const size_t want = bytes + bytes / 8 + 256;  // a little slack so that small growth does not reallocate
            hipError_t e = hipMalloc(&c->ws_ptr[k], want);
            if (e != hipSuccess) return e;
            c->ws_bytes[k] = want;
        }
        *p = c->ws_ptr[k];
        return hipSuccess;
    };
#define BCHK(call)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            free_scene(c);                                                                      \
            return fail(c, RT_EDEVICE, "%s: %s", #call, hipGetErrorString(e_));                 \
        }                                                                                       \
    } while (0)
    uint32_t* d_root = nullptr;
    BCHK(dalloc((void**)&d_root, 12 * sizeof(uint32_t)));
    BCHK(dalloc((void**)&s.tcen, (size_t)nf * sizeof(float4)));
    BCHK(dalloc((void**)&s.tmin, (size_t)nf * sizeof(float4)));
    BCHK(dalloc((void**)&s.tmax, (size_t)nf * sizeof(float4)));
    for (int k = 0; k < 2; k++) {
        BCHK(dalloc((void**)&s.order[k], (size_t)nf * sizeof(uint32_t)));
        BCHK(dalloc((void**)&s.node_of[k], (size_t)nf * sizeof(int32_t)));
        BCHK(dalloc((void**)&s.active[k], max_active * sizeof(int32_t)));
    }
    BCHK(dalloc((void**)&s.nmin, (size_t)max_nodes * sizeof(float4)));
    BCHK(dalloc((void**)&s.nmax, (size_t)max_nodes * sizeof(float4)));
    BCHK(dalloc((void**)&s.cmin, (size_t)max_nodes * sizeof(float4)));
    BCHK(dalloc((void**)&s.cmax, (size_t)max_nodes * sizeof(float4)));
    int32_t** nints[] = {&s.nstart, &s.ncount, &s.nchild, &s.ndepth, &s.nrec, &s.nslot, &s.nsplit, &s.cursors};
    for (auto pp : nints) BCHK(dalloc((void**)pp, (size_t)max_nodes * sizeof(int32_t)));
    BCHK(dalloc((void**)&s.bins, max_active * BB_NODE_WORDS * sizeof(uint32_t)));
    BCHK(dalloc((void**)&s.counters, 8 * sizeof(int32_t)));
    BCHK(dalloc((void**)&s.nparent, (size_t)max_nodes * sizeof(int32_t)));
    BCHK(dalloc((void**)&s.nflag, (size_t)max_nodes * sizeof(int32_t)));
    BCHK(dalloc((void**)&s.small, ((size_t)nf / 2 + 2) * sizeof(int32_t)));
    BCHK(dalloc((void**)&s.scratch, (size_t)nf * 4 * sizeof(float4)));

    // output buffers, sized for the worst case (nf - 1 inner nodes) and kept while the face count fits
    if (bd.out_cap < nf || !c->d_nodes || !c->d_tris || !c->d_shade) {
        free_scene(c);
        BCHK(hipMalloc(&c->d_nodes, (size_t)nf * 4 * sizeof(float4)));
        BCHK(hipMalloc(&c->d_tris, (size_t)nf * 4 * sizeof(float4)));
        BCHK(hipMalloc(&c->d_shade, (size_t)nf * 2 * sizeof(float4)));
        bd.out_cap = nf;
    }
    c->has_scene = false;
    c->phits_valid = false;
    bd.ref_valid = false;

    const uint32_t root_init[12] = {BB_KEY_POS_INF, BB_KEY_POS_INF, BB_KEY_POS_INF, BB_KEY_NEG_INF, BB_KEY_NEG_INF, BB_KEY_NEG_INF,
                                    BB_KEY_POS_INF, BB_KEY_POS_INF, BB_KEY_POS_INF, BB_KEY_NEG_INF, BB_KEY_NEG_INF, BB_KEY_NEG_INF};
    BCHK(hipMemcpyAsync(d_root, root_init, sizeof root_init, hipMemcpyHostToDevice, st));
    BCHK(hipStreamSynchronize(st));

    if (!c->build_ev0) {
        BCHK(hipEventCreate(&c->build_ev0));
        BCHK(hipEventCreate(&c->build_ev1));
    }
    BCHK(hipEventRecord(c->build_ev0, st));
    const dim3 blk(BB_BLOCK), grid_t((nf + BB_BLOCK - 1) / BB_BLOCK);
    // positions per workgroup in the level loop: enough workgroups to fill the chip, at most BB_MAX_SPAN
    int span = ((nf / 2048) / BB_BLOCK) * BB_BLOCK;
    span = span < BB_BLOCK ? BB_BLOCK : (span > BB_MAX_SPAN ? BB_MAX_SPAN : span);
    const dim3 grid_s((nf + span - 1) / span);
    hipLaunchKernelGGL(k_build_prep, grid_t, blk, 0, st, bd.d_verts, bd.d_faces, nf, s, d_root);
    hipLaunchKernelGGL(k_build_root, dim3(1), dim3(64), 0, st, s, d_root, nf);
    int cur = 0, cur_list = 0, levels = 0;
    int32_t h_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        BCHK(hipMemcpyAsync(h_counters, s.counters, sizeof h_counters, hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        const int active = h_counters[2];
        if (active <= 0 || levels > BB_MAX_DEPTH + 1) break;
        if ((size_t)active > max_active) {
            free_scene(c);
            return fail(c, RT_EDEVICE, "builder: %d active nodes exceed the workspace for %zu", active, max_active);
        }
        const size_t words = (size_t)active * BB_NODE_WORDS;
        hipLaunchKernelGGL(k_build_clear_bins, dim3((unsigned)((words + BB_BLOCK - 1) / BB_BLOCK)), blk, 0, st, s.bins, words);
        BCHK(hipMemsetAsync(s.counters + 2, 0, sizeof(int32_t), st));
        if (v1)
            hipLaunchKernelGGL(k_build_bin, grid_t, blk, 0, st, s, cur, nf);
        else
            hipLaunchKernelGGL(k_build_bin2, grid_s, blk, 0, st, s, cur, nf, span);
        hipLaunchKernelGGL(k_build_split, dim3((active + BB_BLOCK - 1) / BB_BLOCK), blk, 0, st, s, cur_list, active, max_nodes, small_max,
                           BS_MIN_DEPTH);
        if (v1)
            hipLaunchKernelGGL(k_build_partition, grid_t, blk, 0, st, s, cur, nf);
        else
            hipLaunchKernelGGL(k_build_partition2, grid_s, blk, 0, st, s, cur, nf, span);
        BCHK(hipGetLastError());
        cur ^= 1;
        cur_list ^= 1;
        levels++;
    }
    if (h_counters[4] > 0) {
        const dim3 gs(h_counters[4]), bs(64);
        const int ns = h_counters[4], tiny = c->build_tiny;
#define SMALL(M_, T_) hipLaunchKernelGGL((k_build_small<M_, T_>), gs, bs, 0, st, s, cur, ns)
        if (small_max == 128) {
            if (tiny == 0) SMALL(128, 0); else if (tiny == 4) SMALL(128, 4); else SMALL(128, 8);
        } else {
            if (tiny == 0) SMALL(256, 0); else if (tiny == 4) SMALL(256, 4); else SMALL(256, 8);
        }
#undef SMALL
        BCHK(hipGetLastError());
        BCHK(hipMemcpyAsync(h_counters, s.counters, sizeof h_counters, hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
    }
    const int num_nodes = h_counters[0], num_recs = h_counters[1], depth = h_counters[3];
    if (num_nodes > max_nodes || num_recs > nf) {
        free_scene(c);
        return fail(c, RT_EDEVICE, "builder: %d nodes / %d records for %d faces", num_nodes, num_recs, nf);
    }
    hipLaunchKernelGGL(k_build_emit_recs, dim3((num_nodes + BB_BLOCK - 1) / BB_BLOCK), blk, 0, st, s, num_nodes, c->d_nodes);
    uint32_t* const d_edge_max = (uint32_t*)(s.counters + 6);  // counters 6, 7: free after the level loop
    BCHK(hipMemsetAsync(d_edge_max, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_build_emit_tris, grid_t, blk, 0, st, s, cur, nf, bd.d_verts, bd.d_faces, bd.d_fcol, c->d_tris, c->d_shade,
                       d_edge_max);
    BCHK(hipEventRecord(c->build_ev1, st));
    float4 rlo, rhi;
    float edge_max[2];
    BCHK(hipMemcpyAsync(&rlo, s.nmin, sizeof rlo, hipMemcpyDeviceToHost, st));
    BCHK(hipMemcpyAsync(&rhi, s.nmax, sizeof rhi, hipMemcpyDeviceToHost, st));
    BCHK(hipMemcpyAsync(edge_max, d_edge_max, sizeof edge_max, hipMemcpyDeviceToHost, st));
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&c->last_build_ms, c->build_ev0, c->build_ev1);
#undef BCHK

    bd.cur = cur;
    bd.parents_valid = false;
    c->num_recs = (size_t)num_recs;
    c->num_tri_recs = (size_t)nf;
    c->sc.nodes = c->d_nodes;
    c->sc.tris = c->d_tris;
    c->sc.shade = c->d_shade;
    c->sc.rmin[0] = rlo.x; c->sc.rmin[1] = rlo.y; c->sc.rmin[2] = rlo.z;
    c->sc.rmax[0] = rhi.x; c->sc.rmax[1] = rhi.y; c->sc.rmax[2] = rhi.z;
    c->cull = built_cull_pad(rlo, rhi, edge_max, c->tune_cull_reach);
    c->finite_boxes = std::isfinite(edge_max[0]) && std::isfinite(edge_max[1]) && std::isfinite(rlo.x + rlo.y + rlo.z + rhi.x + rhi.y + rhi.z);
    c->sc.root_ref = num_nodes == 1 ? ~0 : 0;  // a single leaf starts at triangle 0; otherwise the root owns record 0
    c->sc.num_tris = nf;
    c->num_tris = nf;
    c->max_depth = depth;
    c->cache_recs = num_recs < RT_BFS_PREFIX ? num_recs : RT_BFS_PREFIX;  // the first records are the top levels
    c->built_nodes = num_nodes;
    c->has_scene = true;
    if (c->max_depth > RT_LDS_STACK && (size_t)(c->max_depth - RT_LDS_STACK + 1) > c->spill_entries) free_buffers(c);
    return RT_OK;
}

int rt_build_scene_device(rt_context* c, const float* vertices, int32_t num_vertices, const uint32_t* faces,
                          int32_t num_faces, const float* face_colours) {
    if (!c) return RT_EINVAL;
    if (!vertices || !faces || !face_colours) return fail(c, RT_EINVAL, "NULL argument");
    if (num_vertices < 1 || num_faces < 1) return fail(c, RT_EINVAL, "need at least one vertex and one face");
    if ((int64_t)num_faces >= (int64_t)(1 << 26)) return fail(c, RT_ELIMIT, "more than 2^26 - 1 faces");
    for (int64_t i = 0; i < (int64_t)num_faces * 3; i++)
        if (faces[i] >= (uint32_t)num_vertices) return fail(c, RT_EINVAL, "face %lld references vertex %u of %d", (long long)(i / 3), faces[i], num_vertices);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BuiltData& bd = c->bd;
    if (bd.nv_cap < num_vertices) {
        dfree(bd.d_verts);
        bd.nv_cap = 0;
        HIPCHK(c, hipMalloc(&bd.d_verts, (size_t)num_vertices * 3 * sizeof(float)));
        bd.nv_cap = num_vertices;
    }
    if (bd.nf_cap < num_faces) {
        dfree(bd.d_faces);
        dfree(bd.d_fcol);
        bd.nf_cap = 0;
        HIPCHK(c, hipMalloc(&bd.d_faces, (size_t)num_faces * 3 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&bd.d_fcol, (size_t)num_faces * sizeof(float4)));
        bd.nf_cap = num_faces;
    }
    bd.nv = num_vertices;
    bd.nf = num_faces;
    HIPCHK(c, hipMemcpyAsync(bd.d_verts, vertices, (size_t)num_vertices * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(bd.d_faces, faces, (size_t)num_faces * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(bd.d_fcol, face_colours, (size_t)num_faces * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    return build_core(c);
}

int rt_rebuild_scene_device(rt_context* c, const float* d_vertices) {
    if (!c) return RT_EINVAL;
    if (!d_vertices) return fail(c, RT_EINVAL, "d_vertices is NULL");
    BuiltData& bd = c->bd;
    if (bd.nf <= 0 || !bd.d_faces || !c->built_nodes) return fail(c, RT_ENOSCENE, "rt_rebuild_scene_device needs a scene built by rt_build_scene_device");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(bd.d_verts, d_vertices, (size_t)bd.nv * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return build_core(c);
}

int rt_refit_scene_device(rt_context* c, const float* d_vertices) {
    if (!c) return RT_EINVAL;
    if (!d_vertices) return fail(c, RT_EINVAL, "d_vertices is NULL");
    BuiltData& bd = c->bd;
    if (bd.nf <= 0 || !bd.d_faces || !c->built_nodes || !c->has_scene)
        return fail(c, RT_ENOSCENE, "rt_refit_scene_device needs a scene built by rt_build_scene_device");
    HIPCHK(c, hipSetDevice(c->device));
    ahead_reset(c);
    hipStream_t st = c->stream;
    const int nf = bd.nf, num_nodes = c->built_nodes;
    const dim3 blk(BB_BLOCK), grid_n((num_nodes + BB_BLOCK - 1) / BB_BLOCK), grid_t((nf + BB_BLOCK - 1) / BB_BLOCK);
    HIPCHK(c, hipMemcpyAsync(bd.d_verts, d_vertices, (size_t)bd.nv * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipEventRecord(c->build_ev0, st));
    if (!bd.parents_valid) {
        hipLaunchKernelGGL(k_build_parents, grid_n, blk, 0, st, bd.s, num_nodes);
        bd.parents_valid = true;
    }
    HIPCHK(c, hipMemsetAsync(bd.s.nflag, 0, (size_t)num_nodes * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_build_refit, grid_n, blk, 0, st, bd.s, bd.cur, num_nodes, bd.d_verts, bd.d_faces);
    hipLaunchKernelGGL(k_build_emit_recs, grid_n, blk, 0, st, bd.s, num_nodes, c->d_nodes);
    uint32_t* const d_edge_max = (uint32_t*)(bd.s.counters + 6);
    HIPCHK(c, hipMemsetAsync(d_edge_max, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_build_emit_tris, grid_t, blk, 0, st, bd.s, bd.cur, nf, bd.d_verts, bd.d_faces, bd.d_fcol, c->d_tris, c->d_shade,
                       d_edge_max);
    HIPCHK(c, hipEventRecord(c->build_ev1, st));
    float4 rlo, rhi;
    float edge_max[2];
    HIPCHK(c, hipMemcpyAsync(&rlo, bd.s.nmin, sizeof rlo, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&rhi, bd.s.nmax, sizeof rhi, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(edge_max, d_edge_max, sizeof edge_max, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&c->last_build_ms, c->build_ev0, c->build_ev1);
    bd.ref_valid = false;
    c->phits_valid = false;
    c->cull = built_cull_pad(rlo, rhi, edge_max, c->tune_cull_reach);
    c->finite_boxes = std::isfinite(edge_max[0]) && std::isfinite(edge_max[1]) && std::isfinite(rlo.x + rlo.y + rlo.z + rhi.x + rhi.y + rhi.z);
    c->sc.rmin[0] = rlo.x; c->sc.rmin[1] = rlo.y; c->sc.rmin[2] = rlo.z;
    c->sc.rmax[0] = rhi.x; c->sc.rmax[1] = rhi.y; c->sc.rmax[2] = rhi.z;
    return RT_OK;
}

// The reference-layout copy of a device-built scene is made on the first download after a build.
static int ensure_ref_layout(rt_context* c) {
    BuiltData& bd = c->bd;
    if (bd.ref_valid) return RT_OK;
    const int nf = bd.nf;
    const size_t T = (size_t)nf;
    if (bd.ref_cap < nf) {
        free_ref(c);
        HIPCHK(c, hipMalloc(&c->d_ref_nodes, (size_t)2 * T * 12 * sizeof(float)));
        HIPCHK(c, hipMalloc(&c->d_ref_v0, T * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_ref_v1, T * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_ref_v2, T * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_ref_col, T * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_ref_src, T * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->d_ref_links, (size_t)2 * T * 2 * sizeof(int32_t)));
        bd.ref_cap = nf;
    }
    const dim3 blk(BB_BLOCK);
    hipLaunchKernelGGL(k_build_emit_ref_nodes, dim3((c->built_nodes + BB_BLOCK - 1) / BB_BLOCK), blk, 0, c->stream, bd.s, c->built_nodes,
                       c->d_ref_nodes, c->d_ref_links);
    hipLaunchKernelGGL(k_build_emit_ref_tris, dim3((nf + BB_BLOCK - 1) / BB_BLOCK), blk, 0, c->stream, bd.s, bd.cur, nf, bd.d_verts,
                       bd.d_faces, bd.d_fcol, c->d_ref_v0, c->d_ref_v1, c->d_ref_v2, c->d_ref_col, c->d_ref_src);
    HIPCHK(c, hipGetLastError());
    bd.ref_valid = true;
    return RT_OK;
}

int rt_get_built_scene(rt_context* c, int32_t* num_nodes, int32_t* num_tris, float* nodes, float* triv0, float* triv1,
                       float* triv2, float* colours, int32_t* tri_src, float* build_ms) {
    if (!c) return RT_EINVAL;
    if (!c->has_scene || !c->built_nodes) return fail(c, RT_ENOSCENE, "the current scene was not built on the device");
    if (num_nodes) *num_nodes = c->built_nodes;
    if (num_tris) *num_tris = c->num_tris;
    if (build_ms) *build_ms = c->last_build_ms;
    if (!nodes && !triv0 && !triv1 && !triv2 && !colours && !tri_src) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_ref_layout(c);
    if (rc != RT_OK) return rc;
    const size_t T = (size_t)c->bd.nf;
    if (nodes) HIPCHK(c, hipMemcpyAsync(nodes, c->d_ref_nodes, (size_t)c->built_nodes * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (triv0) HIPCHK(c, hipMemcpyAsync(triv0, c->d_ref_v0, T * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (triv1) HIPCHK(c, hipMemcpyAsync(triv1, c->d_ref_v1, T * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (triv2) HIPCHK(c, hipMemcpyAsync(triv2, c->d_ref_v2, T * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (colours) HIPCHK(c, hipMemcpyAsync(colours, c->d_ref_col, T * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (tri_src) HIPCHK(c, hipMemcpyAsync(tri_src, c->d_ref_src, T * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_built_links(rt_context* c, int32_t* links) {
    if (!c) return RT_EINVAL;
    if (!links) return fail(c, RT_EINVAL, "links is NULL");
    if (!c->has_scene || !c->built_nodes) return fail(c, RT_ENOSCENE, "the current scene was not built on the device");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_ref_layout(c);
    if (rc != RT_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(links, c->d_ref_links, (size_t)c->built_nodes * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_normals(rt_context* c, float* out) {
    if (!c) return RT_EINVAL;
    if (!out) return fail(c, RT_EINVAL, "out_normals is NULL");
    if (!c->has_scene) return fail(c, RT_ENOSCENE, "no scene uploaded");
    std::vector<float> tmp((size_t)c->num_tris * 8);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_shade, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->num_tris; i++) std::memcpy(out + 4 * (size_t)i, tmp.data() + 8 * (size_t)i, 4 * sizeof(float));
    return RT_OK;
}

int rt_raytrace_frame(rt_context* c, const rt_camera* cam, const rt_params* p, uint32_t frame_count,
                      const float* d_prev_frame, float* d_out_frame) {
    int rc = check_render_args(c, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_out_frame) return fail(c, RT_EINVAL, "d_out_frame is NULL");
    const bool sum = (p->flags & RT_FLAG_SUM) != 0;
    const bool fix = (p->flags & RT_FLAG_FIX_INVDIR) != 0;
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    const bool wf_asked = (p->flags & RT_FLAG_WAVEFRONT) != 0;
    const bool mega = (p->flags & RT_FLAG_MEGAKERNEL) != 0 || p->render_mode != RT_MODE_RENDER || p->ray_bounces == 0 ||
                      (p->rays_per_pixel != 1 && (wf_asked || !rpp_fits_paths(p)));
    const bool wavefront = !mega && wf_asked;
    if (!d_prev_frame && (sum || frame_count != 0)) return fail(c, RT_EINVAL, "d_prev_frame is NULL for frame %u", frame_count);
    HIPCHK(c, hipSetDevice(c->device));
    DCamera dc = make_dcamera(cam, p->width, p->height);
    if (!mega && !wavefront) return frame_call(c, dc, p, frame_count, (const float4*)d_prev_frame, (float4*)d_out_frame);
    ahead_reset(c);
    if ((rc = ensure_buffers(c, p->width, p->height)) != RT_OK) return rc;

    FrameArgs fa;
    fa.frame = frame_count;
    fa.bounces = p->ray_bounces;
    fa.rpp = p->rays_per_pixel;
    fa.mode = p->render_mode;
    fa.sum_mode = sum ? 1u : 0u;
    fa.read_prev = (sum || frame_count != 0) ? 1u : 0u;
    fa.tiles_x = (p->width + 15) / 16;
    fa.num_tiles = fa.tiles_x * ((p->height + 15) / 16);
    const float4* prev = (const float4*)d_prev_frame;
    float4* out = (float4*)d_out_frame;
    unsigned long long* gcnt = count ? c->d_counters : nullptr;
    const size_t sstride = c->cap_slots;
    dim3 blk(RT_BLOCK), grid_px(fa.num_tiles);

    if (mega) {
        LaunchTimer lt(c, RT_K_MEGA);
        const bool need_cnt = count || p->render_mode != RT_MODE_RENDER;
#define MEGA(C_, F_) hipLaunchKernelGGL((k_mega<C_, F_>), grid_px, blk, 0, c->stream, c->sc, dc, fa, prev, out, gcnt, c->d_spill, sstride)
        if (need_cnt) { if (fix) MEGA(true, true); else MEGA(true, false); }
        else { if (fix) MEGA(false, true); else MEGA(false, false); }
#undef MEGA
        HIPCHK(c, hipGetLastError());
        return RT_OK;
    }

    uint32_t* counts = c->d_counts;
    {
        LaunchTimer lt(c, RT_K_GENERATE);
        hipLaunchKernelGGL(k_generate, grid_px, blk, 0, c->stream, dc, fa, c->q[0], counts, gcnt);
    }
    uint32_t wf = p->ray_bounces < RT_WF_BOUNCES ? p->ray_bounces : RT_WF_BOUNCES;
    for (uint32_t b = 0; b < wf; b++) {
        DQueue qin = c->q[b & 1], qout = c->q[(b + 1) & 1];
        {
            LaunchTimer lt(c, RT_K_EXTEND);
            if (count)
                hipLaunchKernelGGL((k_extend<true>), grid_px, blk, 0, c->stream, c->sc, qin, counts + b, c->d_hits, gcnt, c->d_spill, sstride);
            else
                hipLaunchKernelGGL((k_extend<false>), grid_px, blk, 0, c->stream, c->sc, qin, counts + b, c->d_hits, gcnt, c->d_spill, sstride);
        }
        {
            LaunchTimer lt(c, RT_K_SHADE);
            if (fix)
                hipLaunchKernelGGL((k_shade<true>), grid_px, blk, 0, c->stream, c->sc, fa, b, qin, counts + b, c->d_hits, qout, counts + b + 1, prev, out);
            else
                hipLaunchKernelGGL((k_shade<false>), grid_px, blk, 0, c->stream, c->sc, fa, b, qin, counts + b, c->d_hits, qout, counts + b + 1, prev, out);
        }
    }
    if (wf < p->ray_bounces) {
        LaunchTimer lt(c, RT_K_TAIL);
        DQueue q = c->q[wf & 1];
        dim3 grid_tail(RT_TAIL_BLOCKS);
#define TAIL(C_, F_) hipLaunchKernelGGL((k_tail<C_, F_>), grid_tail, blk, 0, c->stream, c->sc, fa, wf, q, counts + wf, prev, out, gcnt, c->d_spill, sstride)
        if (count) { if (fix) TAIL(true, true); else TAIL(true, false); }
        else { if (fix) TAIL(false, true); else TAIL(false, false); }
#undef TAIL
    }
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// rt_raytrace_frame for callers that cannot allocate: same kernels, staging and spill area in the static scratch array.
int rt_raytrace_frame_noalloc(rt_context* c, const rt_camera* cam, const rt_params* p, uint32_t frame_count,
                              const float* d_prev_frame, float* d_out_frame) {
    int rc = check_render_args(c, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_out_frame) return fail(c, RT_EINVAL, "d_out_frame is NULL");
    if ((p->flags & (RT_FLAG_MEGAKERNEL | RT_FLAG_WAVEFRONT)) || p->render_mode != RT_MODE_RENDER || p->rays_per_pixel != 1 ||
        p->ray_bounces == 0)
        return fail(c, RT_EINVAL, "rt_raytrace_frame_noalloc supports the default pipeline only (render mode 1, rays_per_pixel 1, ray_bounces >= 1)");
    if (p->flags & RT_FLAG_REUSE_PRIMARY) return fail(c, RT_EINVAL, "rt_raytrace_frame_noalloc does not support RT_FLAG_REUSE_PRIMARY (its hit buffer is allocated)");
    const bool sum = (p->flags & RT_FLAG_SUM) != 0;
    if (!d_prev_frame && (sum || frame_count != 0)) return fail(c, RT_EINVAL, "d_prev_frame is NULL for frame %u", frame_count);
    // scratch layout: [staged samples of one frame: pixels x 16 B][pad to 256 B][spill: entries x threads x 8 B]
    const size_t pixels = (size_t)p->width * p->height;
    const size_t spill_threads = (size_t)c->num_cus * PT_BLOCKS_PER_CU * PT_BLOCK;
    const size_t spill_entries = c->max_depth > PT_STACK ? (size_t)(c->max_depth - PT_STACK + 1) : 0;
    const size_t samples_bytes = (pixels * sizeof(float4) + 255) & ~(size_t)255;
    const size_t need = samples_bytes + spill_entries * spill_threads * sizeof(uint2);
    if (need > RT_NOALLOC_SCRATCH_BYTES)
        return fail(c, RT_ELIMIT,
                    "rt_raytrace_frame_noalloc: %dx%d pixels with a BVH of depth %d need %zu bytes of scratch (16 per pixel + %zu per "
                    "level beyond %d), the static scratch holds %zu; use rt_raytrace_frame",
                    p->width, p->height, c->max_depth, need, spill_threads * sizeof(uint2), PT_STACK - 1, (size_t)RT_NOALLOC_SCRATCH_BYTES);
    HIPCHK(c, hipSetDevice(c->device));
    const DCamera dc = make_dcamera(cam, p->width, p->height);
    const BandLayout bl = band_layout(p, false);
    PathsArgs a;
    fill_paths_args(c, p, bl, frame_count, 1u, 1u, a);
    a.samples = (float4*)c->scratch_ptr;
    a.spill = spill_entries ? (uint2*)(c->scratch_ptr + samples_bytes) : nullptr;
    // Both kernels go to the context's stream, so consecutive calls on one context run one after the other.  Nothing in
    // the scratch has to start at zero: the path kernel writes the sample of every pixel of the band and every spill
    // entry before anything reads it.
    launch_k_paths(c, dc, a, (p->flags & RT_FLAG_COUNTERS) != 0, (p->flags & RT_FLAG_FIX_INVDIR) != 0, PM_TRACE, false, c->stream);
    HIPCHK(c, hipGetLastError());
    launch_k_accumulate(c, p, bl, a, (const float4*)d_prev_frame, (float4*)d_out_frame);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

int rt_render_device(rt_context* c, const rt_camera* cam, const rt_params* p, float* d_accum) {
    int rc = check_render_args(c, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_accum) return fail(c, RT_EINVAL, "d_accum is NULL");
    uint32_t stride = p->frame_stride ? p->frame_stride : 1u;
    if ((p->flags & RT_FLAG_SUM) && p->frame_base == 0) {
        const BandLayout bl = band_layout(p);  // only the band that will be rendered is cleared
        for (const auto& r : bl.ranges)
            HIPCHK(c, hipMemsetAsync(d_accum + (size_t)r.first * p->width * 4, 0, (size_t)r.second * p->width * 4 * sizeof(float), c->stream));
    }
    const bool per_frame = (p->flags & (RT_FLAG_MEGAKERNEL | RT_FLAG_WAVEFRONT)) != 0 ||
                           p->render_mode != RT_MODE_RENDER || !rpp_fits_paths(p) || p->ray_bounces == 0;
    if (per_frame) {
        for (uint32_t f = 0; f < p->frames; f++) {
            uint32_t fc = p->frame_base + f * stride;
            rc = rt_raytrace_frame(c, cam, p, fc, d_accum, d_accum);
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    DCamera dc = make_dcamera(cam, p->width, p->height);
    const uint32_t batch = paths_batch((size_t)p->width * p->height) / p->rays_per_pixel;  // frames x samples staged per launch
    const uint32_t launches = (p->frames + batch - 1) / batch;
    const uint32_t even = c->tune_split == 1 ? (p->frames + launches - 1) / launches : batch;
    for (uint32_t f = 0, n = 0; f < p->frames; f += n) {
        n = p->frames - f < even ? p->frames - f : even;
        if (c->tune_split == 2 && f == 0 && p->frames > batch && p->frames % batch) n = p->frames % batch;   // remainder first
        // "followed": another launch of this call comes after this one, or this one is a short remainder that starts while
        // its predecessor still drains (workgroups that start late carry a full share of the items: bench.py --steps 20,
        // 16 + 4 frames: 4.70 ms with one workgroup per CU for the remainder, 4.75 with two; 16, 64 steps: 4.43 -> 4.01,
        // 15.6 -> 15.1 ms with two for the last full launch)
        rc = launch_paths_batch(c, dc, p, p->frame_base + f * stride, stride, n, (const float4*)d_accum, (float4*)d_accum, true,
                                f + n < p->frames || (f > 0 && 2u * n <= batch), f > 0);   // joins: its primary pass runs next to the previous launch's path kernel
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_render(rt_context* c, const rt_camera* cam, const rt_params* p, float* out_rgba) {
    int rc = check_render_args(c, cam, p);
    if (rc != RT_OK) return rc;
    if (!out_rgba) return fail(c, RT_EINVAL, "out_rgba is NULL");
    size_t pixels = (size_t)p->width * p->height;
    HIPCHK(c, hipSetDevice(c->device));
    if (pixels > c->frame_pixels) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dfree(c->d_frame);
        HIPCHK(c, hipMalloc(&c->d_frame, pixels * sizeof(float4)));
        c->frame_pixels = pixels;
    }
    // only the band being rendered travels; the rest of the caller's buffer is left alone
    const BandLayout bl = band_layout(p);
    for (const auto& r : bl.ranges) {
        float4* d_rows = c->d_frame + (size_t)r.first * p->width;
        const float* h_rows = out_rgba + (size_t)r.first * p->width * 4;
        const size_t bytes = (size_t)r.second * p->width * sizeof(float4);
        if (p->frame_base != 0)
            HIPCHK(c, hipMemcpyAsync(d_rows, h_rows, bytes, hipMemcpyHostToDevice, c->stream));
        else
            HIPCHK(c, hipMemsetAsync(d_rows, 0, bytes, c->stream));
    }
    rc = rt_render_device(c, cam, p, (float*)c->d_frame);
    if (rc != RT_OK) return rc;
    for (const auto& r : bl.ranges)
        HIPCHK(c, hipMemcpyAsync(out_rgba + (size_t)r.first * p->width * 4, c->d_frame + (size_t)r.first * p->width,
                                 (size_t)r.second * p->width * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_primary_hits(rt_context* c, const rt_camera* cam, int32_t width, int32_t height, int32_t* out_ids, float* out_t) {
    if (!c) return RT_EINVAL;
    if (!cam || !out_ids || !out_t) return fail(c, RT_EINVAL, "NULL argument");
    if (!c->has_scene) return fail(c, RT_ENOSCENE, "no scene uploaded");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)16384 * 16384)
        return fail(c, RT_EINVAL, "bad image size %dx%d", width, height);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_buffers(c, width, height);
    if (rc != RT_OK) return rc;
    size_t pixels = (size_t)width * height;
    DevTmp<int> ids;
    DevTmp<float> ts;
    HIPCHK(c, ids.alloc(pixels));
    HIPCHK(c, ts.alloc(pixels));
    int* const d_ids = ids.p;
    float* const d_t = ts.p;
    DCamera dc = make_dcamera(cam, width, height);
    int tiles_x = (width + 15) / 16, tiles = tiles_x * ((height + 15) / 16);
    hipLaunchKernelGGL(k_primary_hits, dim3(tiles), dim3(RT_BLOCK), 0, c->stream, c->sc, dc, tiles_x, d_ids, d_t,
                       c->d_spill, c->cap_slots);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_ids, d_ids, pixels * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_t, d_t, pixels * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    const hipError_t es = hipStreamSynchronize(c->stream);  // always drain before the temporaries go
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(c, RT_EDEVICE, "rt_primary_hits: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_counters(rt_context* c, uint64_t out[RT_NUM_COUNTERS]) {
    if (!c) return RT_EINVAL;
    if (!out) return fail(c, RT_EINVAL, "out is NULL");
    unsigned long long tmp[RT_NUM_COUNTERS];
    HIPCHK(c, hipMemcpyAsync(tmp, c->d_counters, sizeof tmp, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < RT_NUM_COUNTERS; k++) out[k] = tmp[k];
#ifdef PT_DIAG
    {
        unsigned long long d[12];
        (void)hipMemcpy(d, c->d_counters + 8, sizeof d, hipMemcpyDeviceToHost);
        fprintf(stderr, "PT_DIAG inner iters %llu lanes/iter %.2f (waiting at a leaf %.2f, finished %.2f, empty %.2f) | service iters %llu lanes/iter %.2f | tri iters %llu lanes/iter %.2f\n",
                d[0], d[0] ? (double)d[1] / d[0] : 0.0, d[0] ? (double)d[3] / d[0] : 0.0, d[0] ? (double)d[9] / d[0] : 0.0,
                d[0] ? (double)d[11] / d[0] : 0.0, d[4], d[4] ? (double)d[5] / d[4] : 0.0, d[6], d[6] ? (double)d[7] / d[6] : 0.0);
    }
#endif
    return RT_OK;
}

int rt_reset_counters(rt_context* c) {
    if (!c) return RT_EINVAL;
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 32 * sizeof(unsigned long long), c->stream));
    return RT_OK;
}

int rt_set_profiling(rt_context* c, int enable) {
    if (!c) return RT_EINVAL;
    ahead_reset(c);
    c->profiling = enable != 0;
    return RT_OK;
}

int rt_kernel_times(rt_context* c, float out_ms[RT_NUM_KERNEL_CLASSES], uint32_t out_launches[RT_NUM_KERNEL_CLASSES]) {
    if (!c) return RT_EINVAL;
    if (!out_ms || !out_launches) return fail(c, RT_EINVAL, "NULL argument");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& ep : c->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            c->ms[ep.cls] += ms;
            c->launches[ep.cls]++;
        }
        c->free_events.push_back(ep);
    }
    c->pending.clear();
    for (int k = 0; k < RT_NUM_KERNEL_CLASSES; k++) {
        out_ms[k] = (float)c->ms[k];
        out_launches[k] = c->launches[k];
        c->ms[k] = 0.0;
        c->launches[k] = 0;
    }
    return RT_OK;
}

int rt_selftest_math(rt_context* c, const float* x, int32_t n, float* out) {
    if (!c) return RT_EINVAL;
    if (!x || !out || n <= 0) return fail(c, RT_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp<float> tx, tout;
    HIPCHK(c, tx.alloc((size_t)n));
    HIPCHK(c, tout.alloc((size_t)n * 6));
    float *const dx = tx.p, *const dout = tout.p;
    hipError_t e = hipMemcpyAsync(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, dx, n, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    const hipError_t es = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(c, RT_EDEVICE, "rt_selftest_math: %s", hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
