/*
 * rt_api.h -- C ABI of the MI355X render path (librt_hip.so).
 *
 * This is the drop-in boundary for the reference's render path: the two functions of
 * raytrace.h:9-10 (raytraceInit / raytrace), the SSBO uploads and uniform set-up behind
 * them (raytrace.cpp:38-87), and the image ping-pong the caller performs around each
 * dispatch (main.cpp:189-201).  No GL, no C++ types, no exceptions: plain pointers and
 * sizes in, int status out (0 = RT_OK), message through rt_last_error().
 *
 * Conventions
 *   - A context belongs to one GPU and is used from one host thread at a time.  All work
 *     is issued on the context's stream (its own by default, rt_set_stream() to share the
 *     caller's).  Entry points taking device pointers are asynchronous on that stream;
 *     entry points taking host pointers return after the data is valid.
 *   - Framebuffers are W*H*4 floats, linear RGBA32F, row 0 = BOTTOM row of the picture
 *     (GL image coordinates, raytrace.glsl:282,294), value = running mean over frames,
 *     alpha = (1 + rays_per_pixel) / rays_per_pixel (raytrace.glsl:289-293).
 *   - "frame" = one dispatch of the reference = rays_per_pixel samples per pixel seeded
 *     with pixelIndex + frameCount*745621 (raytrace.glsl:277).  N spp of the shipped
 *     constants (RAYS_PER_PIXEL = 1) = frames 0..N-1.
 *   - Results are bit-identical to the CPU oracle built on include/rt_math.h.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_EINVAL (-1)    /* bad argument or malformed scene */
#define RT_EDEVICE (-2)   /* HIP runtime error (message has the HIP error string) */
#define RT_ENOSCENE (-3)  /* render requested before rt_upload_scene */
#define RT_ELIMIT (-4)    /* size beyond a documented limit */

/* render modes, constants.h:16-19 */
#define RT_MODE_RENDER 1
#define RT_MODE_TRIANGLE_TESTS 2
#define RT_MODE_BOX_TESTS 3
#define RT_MODE_REFLECTIONS 4

/* rt_params.flags */
#define RT_FLAG_FIX_INVDIR 0x1u   /* refresh ray.invDir after each bounce (the reference never does,
                                     raytrace.glsl:247-254,288); OFF = bug-compatible default */
#define RT_FLAG_SUM 0x2u          /* framebuffer receives the plain SUM of the frames' colours instead of
                                     the running mean (for sample-partitioned multi-GPU rendering) */
#define RT_FLAG_MEGAKERNEL 0x4u   /* pipeline: one thread follows one pixel's whole sample loop, one launch per
                                     frame (the reference's dispatch shape) */
#define RT_FLAG_COUNTERS 0x8u     /* also accumulate the work counters read by rt_counters() */
#define RT_FLAG_WAVEFRONT 0x10u   /* pipeline: separate generate / extend / shade+compact kernels over ray
                                     queues in HBM, one frame at a time.  Without a pipeline flag the default
                                     is the persistent path kernel, which renders batches of frames
                                     concurrently.  All pipelines produce identical pixels. */

#define RT_FLAG_NO_CULL 0x40u      /* traverse exactly the nodes the reference traverses.  By default the path kernel skips
                                     the subtree under a box that a bounce ray's TRUE line passes at a distance: with the
                                     stale invDir (raytrace.glsl:247-254) the reference's slab tests follow the primary
                                     ray's line instead, so it walks through subtrees none of whose triangles the ray can
                                     hit.  Skipping them cannot change a pixel (the pad of the test covers the rounding of
                                     the triangle test); it changes the work counters, which count what ran.  With this
                                     flag they equal the reference's numBoxTests / numTriangleTests.  No effect with
                                     RT_FLAG_FIX_INVDIR (the slab test then is the true one) or on the other pipelines */
#define RT_FLAG_REUSE_PRIMARY 0x20u /* OPT-IN shortcut, not what the reference does: the reference shoots the same primary ray
                                     for a pixel in every frame (no jitter, raytrace.glsl:279-288), so while camera, image
                                     size and scene stand still its closest hit is the same.  With this flag the hits are
                                     traced once (kept in the context until camera, size, band or scene change) and every
                                     path starts at its first bounce.  Pixels are bit-identical; rt_counters() then counts the
                                     rays actually traced (the primary pass once + bounce rays).  Default pipeline only. */

#define RT_FLAG_PACKETS 0x80u      /* trace the primary rays tile by tile whatever the library's own estimate says.  By default
                                     the default pipeline decides per launch: where triangles are large against a pixel's
                                     footprint (estimated from the root box, the triangle count and the camera) the 64 primary
                                     rays of every 8x8-pixel tile walk the BVH together, one node fetch per wave, each ray in
                                     the reference's own visit order (raytrace.glsl:127-195); elsewhere every lane follows
                                     its own ray.  Same pixels and counters either way;
                                     the flag exists so that tests and measurements can pin the choice.  Ignored with
                                     RT_FLAG_NO_CULL (the literal lane-by-lane traversal), RT_FLAG_REUSE_PRIMARY and on the
                                     other pipelines.  No counterpart in the reference. */

/* Scene in the layout the reference uploads (raytrace.cpp:38-54, scene-loader.h:7-15):
 *   nodes    num_nodes x 12 floats, AlignedMat3 {min.xyz,_}{max.xyz,_}{isLeaf,a,b,_};
 *            leaf iff |isLeaf-1| <= 1e-4, a..b inclusive triangle range, else child ids
 *            (raytrace.glsl:155-168); node 0 is the root
 *   triv0/1/2  num_tris x 4 floats (xyz + pad), the three vertices of each triangle
 *   colours  num_tris x 4 floats (rgb + pad), the per-triangle albedo of raytrace.cpp:44-52
 *   links    optional (may be NULL): num_nodes x 2 int32 (a, b) that replace the float-encoded a, b of nodes.  A
 *            float holds integers exactly only up to 2^24, so the reference's own encoding (bvh.cpp:177-187) cannot
 *            address larger trees; scenes with more than 2^24 nodes or triangles must pass links.
 * All host pointers; rt_upload_scene copies. */
typedef struct rt_scene {
    int32_t num_nodes;
    int32_t num_tris;
    const float* nodes;
    const float* triv0;
    const float* triv1;
    const float* triv2;
    const float* colours;
    const int32_t* links;
} rt_scene;

/* Per-frame uniforms cameraPos / cameraRotation (raytrace.cpp:80-83) and the static
 * u_FOV / u_ViewportDist (raytrace.cpp:69-71).  rot is column-major (GLM order). */
typedef struct rt_camera {
    float pos[3];
    float rot[9];
    float fov_rad;        /* reference: (float)(90 * pi / 180) */
    float viewport_dist;  /* reference: 0.1 */
} rt_camera;

/* u_ScreenWidth/Height, u_RaysPerPixel, u_RayBounces, renderMode, frameCount
 * (raytrace.cpp:65-73,84-85; defaults constants.h:21-24). */
typedef struct rt_params {
    int32_t width;
    int32_t height;
    uint32_t rays_per_pixel;  /* 1 */
    uint32_t ray_bounces;     /* 100 */
    uint32_t render_mode;     /* RT_MODE_RENDER */
    uint32_t frame_base;      /* frameCount of the first frame rendered */
    uint32_t frames;          /* number of frames (= spp at rays_per_pixel 1) */
    uint32_t frame_stride;    /* frameCount step between frames; 0 or 1 = consecutive */
    uint32_t flags;           /* RT_FLAG_* */
    uint32_t band;            /* 0 = whole image; (index << 16) | count = render only horizontal band `index` of
                                 `count` (the image-tile partition over GPUs, count <= 32767).  Contiguous bands:
                                 band i covers tile rows [i*R/count, (i+1)*R/count), R = ceil(height/8).  With
                                 RT_BAND_INTERLEAVED or-ed in, band i is the 16-pixel-row stripes number i, i+count,
                                 i+2*count, ... (bottom to top), which spreads an object in the middle of the image over
                                 all bands.  Pixels outside the band are neither read nor written; seeds use global pixel
                                 indices, so the union of the bands is bit-identical to the whole image.  Default
                                 pipeline only. */
} rt_params;

typedef struct rt_context rt_context;

/* Number of uint64 words rt_counters() writes:
 * [0] box tests (numBoxTests, raytrace.glsl:114)   [1] triangle tests (numTriangleTests, :80)
 * [2] rays = calls of getHitInfo (:245)            [3] rays that hit a triangle
 * [4] node pops that reached the node fetch (:154) [5] pixel samples (paths) */
#define RT_BAND_INTERLEAVED 0x8000u /* rt_params.band: stripes of 16 pixel rows dealt round-robin to the bands */
#define RT_NUM_COUNTERS 6

/* ---- life cycle -------------------------------------------------------------------- */
/* Creates a context on HIP device `device_id` (the GL context + compiled program of
 * raytraceInit, raytrace.cpp:59-62). */
int rt_create(int device_id, rt_context** out_ctx);
int rt_destroy(rt_context* ctx);
/* Last error message of this context ("" if none); valid until the next call on ctx.
 * With ctx == NULL returns the message of the last failed rt_create of this thread. */
const char* rt_last_error(const rt_context* ctx);
/* Use the caller's hipStream_t (passed as void*) for all subsequent work; NULL restores
 * the context's own stream.  (A NULL handle therefore never means HIP's legacy default stream: a caller that wants the
 * library's work there passes hipStreamLegacy, whose handle is not NULL.) */
int rt_set_stream(rt_context* ctx, void* hip_stream);
/* The hipStream_t the context currently issues its work on (its own unless rt_set_stream changed it), so that a caller
 * can enqueue its own kernels and copies in order with the library's instead of creating another stream. */
int rt_get_stream(rt_context* ctx, void** out_hip_stream);
/* Blocks until all work issued on the context's stream has finished, frames traced ahead included. */
int rt_synchronize(rt_context* ctx);
/* Most frames rt_raytrace_frame traces ahead of a steady frame-by-frame loop in one launch (default 8, at most 16;
 * 0 or 1: never trace ahead -- for callers that move the camera every few frames and want no discarded work).
 * No counterpart in the reference. */
int rt_set_frames_ahead(rt_context* ctx, int32_t max_frames);

/* ---- scene ------------------------------------------------------------------------- */
/* initBuffers (raytrace.cpp:38-57): validates and copies the scene, re-lays it out for the
 * device, and runs the triangle-normal kernel (shaders/normals.glsl:13-19). */
int rt_upload_scene(rt_context* ctx, const rt_scene* scene);
/* Copies the device-computed normals back (num_tris x 4 floats, stride of the std430 vec3[]
 * buffer, raytrace.glsl:33-35). */
int rt_get_normals(rt_context* ctx, float* out_normals);

/* GPU BVH build (no counterpart in the reference, whose builder is host code: bvh.cpp:136-173; this is the
 * "per-frame rebuild" path for animated meshes).  Builds a binned-SAH BVH (16 bins, the reference's cost model
 * bvh.cpp:24-56 and depth cap constants.h:26) over the given triangle soup ON THE DEVICE and makes it the
 * context's scene, as rt_upload_scene would.  vertices: num_vertices x 3 floats, faces: num_faces x 3 indices
 * (0-based), face_colours: num_faces x 4 floats, all host pointers.  Triangle ids reported by rt_primary_hits
 * index the leaf-ordered arrays returned by rt_get_built_scene.  Synchronises. */
int rt_build_scene_device(rt_context* ctx, const float* vertices, int32_t num_vertices, const uint32_t* faces,
                          int32_t num_faces, const float* face_colours);
/* Per-frame rebuild for an animated mesh (no counterpart in the reference): same faces and colours as the last
 * rt_build_scene_device, new vertex positions read from the DEVICE pointer d_vertices (num_vertices x 3 floats, copied
 * on the context's stream).  Reuses every buffer of the previous build: no allocation, no host validation, no upload.
 * Synchronises (the level loop reads one counter pair per level). */
int rt_rebuild_scene_device(rt_context* ctx, const float* d_vertices);
/* Refit instead of rebuild (no counterpart in the reference): keeps the tree topology and the triangle order of the
 * last build and recomputes every box bottom-up from the new vertex positions (DEVICE pointer, as above).  About ten
 * times cheaper than a rebuild; the tree stays valid for any deformation but its quality decays with large ones, so
 * callers interleave rebuilds.  build_ms of rt_get_built_scene reports the refit time.  Synchronises. */
int rt_refit_scene_device(rt_context* ctx, const float* d_vertices);
/* Downloads the device-built scene in the layout of rt_scene (so that any other implementation of the path can
 * render the very same tree): nodes num_nodes x 12, triv0/1/2 and colours num_tris x 4 in leaf order, tri_src
 * = face index of each leaf-ordered triangle, build_ms = device time of the build.  Any pointer may be NULL;
 * call once with only the two counts to size the arrays. */
int rt_get_built_scene(rt_context* ctx, int32_t* num_nodes, int32_t* num_tris, float* nodes, float* triv0, float* triv1,
                       float* triv2, float* colours, int32_t* tri_src, float* build_ms);
/* The exact integer (a, b) of every node of the device-built scene (num_nodes x 2 int32), i.e. rt_scene.links: the
 * float-encoded ids in `nodes` are inexact beyond 2^24. */
int rt_get_built_links(rt_context* ctx, int32_t* links);

/* ---- rendering --------------------------------------------------------------------- */
/* One dispatch of the reference: raytrace(cameraPos, cameraRotation, renderMode, frameCount,
 * ...) (raytrace.h:10) with image unit 6 = d_prev_frame (read) and unit 5 = d_out_frame
 * (written) (main.cpp:189-192).  Both are DEVICE pointers to W*H*4 floats and may alias.
 * d_prev_frame may be NULL when frame_count == 0 (treated as zeros).  Uses params->width,
 * height, rays_per_pixel, ray_bounces, render_mode, flags; asynchronous.
 * Consecutive calls overlap on the GPU: the traversal of a frame runs on an internal stream and only the
 * step that reads d_prev_frame and writes d_out_frame is ordered on the context's stream, which is all a
 * caller can observe -- work enqueued on that stream after the call sees the finished frame.
 * A steady loop (the same camera and params, frame_count going up by one, the reference's loop while the camera
 * stands still, main.cpp:183-203) is recognised after two calls: the coming frames are then traced ahead in batches
 * of up to rt_set_frames_ahead() frames and a call only folds its own frame's samples in.  The pixels are the same;
 * whatever was traced ahead and is not asked for next is discarded. */
int rt_raytrace_frame(rt_context* ctx, const rt_camera* cam, const rt_params* params,
                      uint32_t frame_count, const float* d_prev_frame, float* d_out_frame);

/* rt_raytrace_frame for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same kernels and
 * bit-identical pixels, but it calls no hipMalloc / hipFree: the staged samples of the frame and the traversal-stack
 * spill area live in a static device array of RT_NOALLOC_SCRATCH_BYTES (128 MiB) that every GPU the library runs
 * on holds until the process ends.  Limits: default pipeline only (render mode 1, rays_per_pixel 1, ray_bounces >= 1,
 * no RT_FLAG_MEGAKERNEL / RT_FLAG_WAVEFRONT), and
 *     width*height*16 + (depth > 8 ? depth - 7 : 0) * 8 * (2048 * compute units)  <=  RT_NOALLOC_SCRATCH_BYTES
 * with depth = the deepest level of the BVH, root = 0 (on an MI355X, 256 compute units, 1920x1080 fits with trees up
 * to depth 31, 1280x720 up to depth 35); larger requests fail with RT_ELIMIT.  Both kernels
 * run on the context's stream, so calls on one context are ordered; all contexts of one GPU share that GPU's array,
 * so calls of DIFFERENT contexts (or processes' threads) that overlap on one GPU race -- serialise them. */
#define RT_NOALLOC_SCRATCH_BYTES ((size_t)128 << 20)
int rt_raytrace_frame_noalloc(rt_context* ctx, const rt_camera* cam, const rt_params* params,
                              uint32_t frame_count, const float* d_prev_frame, float* d_out_frame);

/* render(scene, camera, spp) -> framebuffer: params->frames consecutive dispatches starting
 * at frame_base, accumulated in place in d_accum (DEVICE pointer, W*H*4 floats).  With
 * frame_base == 0 the previous content of d_accum is ignored; otherwise it must hold the
 * running mean (or, with RT_FLAG_SUM, the sum) of the earlier frames.  Asynchronous. */
int rt_render_device(rt_context* ctx, const rt_camera* cam, const rt_params* params, float* d_accum);

/* Same with a HOST framebuffer: uploads it when frame_base != 0, renders, downloads,
 * synchronises. */
int rt_render(rt_context* ctx, const rt_camera* cam, const rt_params* params, float* out_rgba);

/* Parity hook: closest hit of every pixel's primary ray.  ids = index into the uploaded
 * triangle arrays or -1, t = hit distance or +inf.  HOST pointers, width*height entries,
 * row 0 = bottom. */
int rt_primary_hits(rt_context* ctx, const rt_camera* cam, int32_t width, int32_t height,
                    int32_t* out_ids, float* out_t);

/* Work counters accumulated by renders issued with RT_FLAG_COUNTERS since the last
 * rt_reset_counters(); synchronises the stream. */
int rt_counters(rt_context* ctx, uint64_t out[RT_NUM_COUNTERS]);
int rt_reset_counters(rt_context* ctx);

/* ---- instrumentation --------------------------------------------------------------- */
/* Kernel classes reported by rt_kernel_times(). */
#define RT_K_GENERATE 0
#define RT_K_EXTEND 1
#define RT_K_SHADE 2
#define RT_K_TAIL 3
#define RT_K_ACCUMULATE 4
#define RT_K_MEGA 5
#define RT_K_PATHS 6
#define RT_K_PRIMARY 7   /* primary rays traced tile by tile ahead of the path kernel */
#define RT_NUM_KERNEL_CLASSES 8
/* enable != 0: bracket every kernel launch with HIP events on the context's stream (adds
 * event overhead; keep it off in throughput runs).  rt_kernel_times synchronises and
 * returns the summed milliseconds and launch counts per class since profiling was enabled
 * or last read, then clears them.  Launches of rt_raytrace_frame_noalloc are not bracketed
 * (it creates no events), so they do not appear here. */
int rt_set_profiling(rt_context* ctx, int enable);
int rt_kernel_times(rt_context* ctx, float out_ms[RT_NUM_KERNEL_CLASSES],
                    uint32_t out_launches[RT_NUM_KERNEL_CLASSES]);

/* Evaluates the shared math of include/rt_math.h ON THE DEVICE for n inputs (host pointers):
 * out[0..n) = log(x), out[n..2n) = cos(x), out[2n..3n) = exp(x), out[3n..4n) = sqrt(x),
 * out[4n..5n) = 1/x, out[5n..6n) = pow(x, 0.35).  Lets tests compare device and host bit for bit. */
int rt_selftest_math(rt_context* ctx, const float* x, int32_t n, float* out);

/* Library build identification: "rt_hip gfx950 <hash of the sources it was built from> <date>". */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
