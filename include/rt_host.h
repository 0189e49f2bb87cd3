/*
 * rt_host.h -- C ABI of the host-side scene ingest (librt_host.so, plain C++,
 * no GPU dependency).  It produces the inputs of the render path in exactly
 * the byte layout the reference's loader hands to its GL upload code, so a
 * scene built here can be passed unchanged to rt_upload_scene() (rt_api.h).
 *
 * Reference interfaces replaced:
 *   readObjContents()                obj-reader.h:13   (obj-reader.cpp:13-40)
 *   generateBVH() / serialiseBVH()   bvh.h:27-29       (bvh.cpp:24-200)
 *   loadScene() -> SceneData*        scene-loader.h:11-17 (scene-loader.cpp:6-30)
 *   random triangle colours          raytrace.cpp:44-52 (made deterministic: explicit seed)
 *
 * All functions return 0 on success and a negative RT_HOST_E* code otherwise;
 * nothing throws across this boundary.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_HOST_OK 0
#define RT_HOST_EIO (-1)      /* file could not be opened (reference: uncaught std::runtime_error) */
#define RT_HOST_EINVAL (-2)   /* bad argument / face index out of range */
#define RT_HOST_EEMPTY (-3)   /* scene without triangles */

/* Constants of the reference builder, constants.h:26-27 */
#define RT_MAX_BVH_DEPTH 32
#define RT_BVH_SPLIT_ITERATIONS 64

/* Mirror of SceneData (scene-loader.h:11-15) with plain arrays.
 *   nodes  : num_nodes x 12 floats = AlignedMat3 {min.xyz,0}{max.xyz,0}{isLeaf,a,b,0}
 *            (scene-loader.h:7-9, bvh.cpp:177-187); node ids are preorder positions.
 *   triv0/1/2 : num_tris x 4 floats {x,y,z,0}, in BVH leaf order (scene-loader.cpp:22-27)
 *   tri_src   : num_tris ints, position of each BVH-ordered triangle in the OBJ's face list
 *               (not in the reference's SceneData; provided for id mapping)
 * All arrays are owned by the rt_host_scene and freed by rt_host_free_scene(). */
typedef struct rt_host_scene {
    int32_t num_tris;
    int32_t num_nodes;
    int32_t num_vertices;
    int32_t num_leaves;
    int32_t min_leaf_size;
    int32_t max_leaf_size;
    int32_t max_depth;      /* depth of the deepest node, root = 0 */
    int32_t reserved;
    float* nodes;
    float* triv0;
    float* triv1;
    float* triv2;
    int32_t* tri_src;
    int32_t* links;         /* num_nodes x 2: the (a, b) of every node as exact integers, taken from the builder's tree and
                               not from the float serialisation (bvh.cpp:177-187), which is exact only up to 2^24 */
} rt_host_scene;

/* OBJ reader with the reference's grammar (obj-reader.cpp:22-38): lines "v x y z",
 * "f a b c" and "f a//n b//n c//n"; 1-based indices; everything else ignored.
 * Outputs are malloc'ed: vertices (nv x 3 floats), faces (nf x 3 uint32, 0-based). */
int rt_host_read_obj(const char* path, float** vertices, int32_t* num_vertices, uint32_t** faces,
                     int32_t* num_faces);
void rt_host_free(void* p);

/* SAH builder + serialisation + alignment = loadScene() minus the file read.
 * Triangle order, node order and every float are identical to what the reference's
 * objects produce when compiled with g++/libstdc++ (including the integer abs() in the
 * ternary-search guard, bvh.cpp:63). */
int rt_host_build_scene(const float* vertices, int32_t num_vertices, const uint32_t* faces,
                        int32_t num_faces, rt_host_scene* out);

/* loadScene() for an explicit path (the reference compiles the path in, constants.h:14). */
int rt_host_load_scene(const char* obj_path, rt_host_scene* out);
void rt_host_free_scene(rt_host_scene* s);

/* Triangle colours as raytrace.cpp:44-52 draws them -- mt19937 +
 * uniform_real_distribution<float>(0,1), three draws per triangle (g++ order: b,g,r), alpha 1 -- from an
 * explicit seed instead of std::random_device.  out: num_tris x 4 floats. */
void rt_host_make_colours(uint32_t seed, int32_t num_tris, float* out);

/* Camera rotation as main.cpp:41-51 builds it: yRotation(yaw) * xRotation(pitch), GLM
 * column-major, trigonometry in double then narrowed to float.  out: 9 floats. */
void rt_host_camera_rotation(double pitch, double yaw, float out[9]);

/* FNV-1a 64-bit hash of a byte range (used for the committed ingest fixtures). */
uint64_t rt_host_hash64(const void* data, uint64_t num_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
