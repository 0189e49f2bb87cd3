// Primary rays traced wave-wide: the 64 primary rays of an 8x8-pixel tile walk the BVH as ONE packet.
//
// The lanes of the path kernel (rt_paths.h) each follow their own ray: own node reference, own stack in LDS, own
// 64-byte record fetch per visit.  The primary rays of a tile are neighbours, though: they leave one point and visit
// nearly the same nodes (dragon80k at 1080p: 15 node visits per ray, 26 for the union over a tile).  k_primary gives a
// tile to a wave, keeps the node reference and the stack in scalar registers, fetches every record ONCE per wave through
// the scalar cache, and every lane tests its own ray against the shared boxes and triangles under a lane mask.
//
// Every lane sees exactly the node sequence the reference's traversal (raytrace.glsl:127-195) gives its ray:
//   * a lane takes part in a node only if it would have visited it: its own distance to the child box is below its own
//     closest hit when the node is entered (raytrace.glsl:174-189) and again when it is taken off the stack (:151);
//   * the reference visits the nearer child first (equal distances: child 2).  The wave goes to the child that most of
//     the lanes wanting BOTH children prefer; the lanes that prefer the other order skip it, take the second child with
//     everybody else and come back for the first one afterwards (an extra stack entry that names only them).
// Per lane the order of triangle tests, the strict `<` updates and every cull are therefore the reference's, and so are
// the closest hit, ties included, and the work counters.
//
// Output per work item (tile, frame):
//   hcount[item]        number of lanes whose ray hit something; PK_TRACE_ITEM if the packet gave up (stack deeper than
//                       PK_STACK entries) -- the path kernel then traces that tile's primary rays lane by lane
//   hitq[item*64 + k]   (t, triangle id, lane, 0) of the k-th hit lane: where the path kernel picks up the path
//   samples[frame][pixel] = (1, 1, 1, dir.y) for every ray that left the scene: a finished path (raytrace.glsl:263-266)
#ifndef RT_PACKETS_H
#define RT_PACKETS_H

#include "rt_paths.h"

#define PK_BLOCK 256                 // 4 waves per workgroup
#define PK_BLOCKS_PER_CU 8           // 32 waves per CU, 8 per SIMD (needs <= 64 VGPRs and 20 KiB of LDS per workgroup)
#ifndef PK_STACK
#define PK_STACK 19                  // packet-stack entries (a 256-byte row of per-lane distances each, in LDS)
#endif
#define PK_TRACE_ITEM 0xffffffffu
#define PK_COUNTERS 64u              // item counters of a k_primary launch, 16 words apart

// How many of them a grid of this size uses, and with how many waves each starts (see k_primary).
__host__ __device__ inline uint32_t pk_counters(uint32_t grid) { return grid % PK_COUNTERS == 0u ? PK_COUNTERS : grid % 8u == 0u ? 8u : 1u; }

typedef float pk_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) pk_f4* pk_cf4;   // constant address space: uniform addresses load through the scalar cache


// The primary rays of a launch leave one point, so the first half of every slab test -- box corner minus ray origin
// (raytrace.glsl:115-116) -- is the same for all of them: k_rebase does those subtractions once per launch, into a copy of the
// inner-node records that k_primary<false> reads (12 vector instructions of a packet's node visit).  The same
// binary32 subtraction, done earlier: results are bit-identical.
__global__ __launch_bounds__(256) void k_rebase(const float4* nodes, float4* out, uint32_t num_quads, float ox, float oy, float oz) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= num_quads) return;
    float4 q = nodes[i];
    switch (i & 3u) {
        case 0: q = make_float4(q.x - ox, q.y - oy, q.z - oz, q.w - ox); break;   // c1.min.xyz c1.max.x
        case 1: q = make_float4(q.x - oy, q.y - oz, q.z - ox, q.w - oy); break;   // c1.max.yz  c2.min.xy
        case 2: q = make_float4(q.x - oz, q.y - ox, q.z - oy, q.w - oz); break;   // c2.min.z   c2.max.xyz
        default: break;                                                          // child references
    }
    out[i] = q;
}

// The packet's whole traversal, hand-written: the step is bound by the scalar unit (one scalar instruction per four cycles
// and SIMD), and the compiler's version of the ordering and stack logic took ~125 scalar and branch instructions per node
// where ~30 do.  Same algorithm, statement for statement, as the loop in k_primary<true> below (which the counting build
// keeps): EXEC is the set of lanes taking part in the current node, the node reference, the stack pointer and the lane
// masks live in SGPRs, the packet's stack in three VGPRs (lane i = entry i, v_writelane / v_readlane) plus one LDS row of
// per-lane distances per entry.  The slab test, the triangle test (with the correctly rounded division) and their
// instruction order are the compiler's own output for box_dist / tri_dist_flat, so results are bit-identical.
// `nodes` are the launch's rebased records (k_rebase: box corners minus the common ray origin); the triangle test still
// needs the origin.  Registers s36-s73 and v40-v61 are used by name: with the compiler's own s0-s35, VCC and the reserved pairs that
// is 80 SGPRs per wave.  A SIMD has 800 and the hardware sets 16 more per wave aside for the trap handler, so 80 is the most that
// leaves room for 8 waves (a version that named registers up to s89 ran 7 per SIMD, and 3 instead of 4 next to a workgroup of
// the path kernel).  Returns true when the stack would have outgrown PK_STACK.
// The text comes in three pieces -- up to the arrival of an inner node's record, the two slab tests (which leave tNear
// of child 1 / child 2 in v51 / v52 and tFar in v50 / v54, raytrace.glsl:117-120), and everything after them -- and PK_DEFINE_TRAVERSE makes one function per form of the slab tests.
// The same for the triangle test (raytrace.glsl:79-110): with one origin for all rays of the launch, s = o - a,
// q = cross(s, e1) and dot(e2, q) -- the numerator of t -- are per-triangle constants.  k_rebase_tris computes them once
// per launch, in binary32 with the operations and their order of tri_dist_flat, into a copy of the triangle records laid out
// (e1 flags)(e2 id)(s, dot(e2, q))(q -): 17 of the 54 vector instructions of a packet's triangle test, and one 64-byte
// scalar load instead of two loads.
__global__ __launch_bounds__(256) void k_rebase_tris(const float4* tris, float4* out, uint32_t num_tris, float ox, float oy, float oz) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= num_tris) return;
    const float4 t0 = tris[(size_t)i * 4u], t1 = tris[(size_t)i * 4u + 1u], t2 = tris[(size_t)i * 4u + 2u];
    const rt_vec3 e1 = rt_v3(t1.x, t1.y, t1.z), e2 = rt_v3(t2.x, t2.y, t2.z);
    const rt_vec3 s = rt_sub(rt_v3(ox, oy, oz), rt_v3(t0.x, t0.y, t0.z));
    const rt_vec3 q = rt_cross(s, e1);
    out[(size_t)i * 4u] = make_float4(e1.x, e1.y, e1.z, t0.w);
    out[(size_t)i * 4u + 1u] = make_float4(e2.x, e2.y, e2.z, t1.w);
    out[(size_t)i * 4u + 2u] = make_float4(s.x, s.y, s.z, rt_dot(e2, q));
    out[(size_t)i * 4u + 3u] = make_float4(q.x, q.y, q.z, 0.0f);
}

#define PK_STR2(x) #x
#define PK_STR(x) PK_STR2(x)
// diagnostic builds: PK_PAD_V vector / PK_PAD_S scalar no-ops per inner-node visit, to measure what an instruction there costs
#ifdef PK_PAD_V
#define PK_ASM_PAD_V ".rept " PK_STR(PK_PAD_V) "\n\tv_nop\n\t.endr\n\t"
#else
#define PK_ASM_PAD_V ""
#endif
#ifdef PK_PAD_S
#define PK_ASM_PAD_S ".rept " PK_STR(PK_PAD_S) "\n\ts_nop 0\n\t.endr\n\t"
#else
#define PK_ASM_PAD_S ""
#endif

#define PK_ASM_HEAD \
        "s_mov_b64 s[72:73], exec\n\t"                                                                                 \
        "v_mov_b32_e32 v53, 0x38d1b717\n\t" /* RT_EPS */                                                                   \
        "v_mov_b32_e32 v60, 0x7f800000\n\t" /* +inf */                                                                 \
        "s_mov_b32 s69, 0\n\t" /* stack pointer */                                                                     \
        "s_mov_b32 %[gave], 0\n\t"                                                                                     \
        "s_mov_b32 s68, %[root]\n\t" /* current node */                                                                \
        "s_and_b64 exec, s[72:73], %[in0]\n\t"                                                                         \
        "s_cbranch_execz Ldone%=\n"                                                                                    \
        "Lnode%=:\n\t"                                                                                                 \
        "s_cmp_lt_i32 s68, 0\n\t"                                                                                      \
        "s_cbranch_scc1 Lleaf%=\n\t"                                                                                   \
        /* ---- inner node: record = (c1.min c1.max c2.min c2.max ref1 ref2 - -) */                                    \
        "s_lshl_b32 s70, s68, 6\n\t"                                                                                   \
        "s_load_dwordx16 s[36:51], %[nodes], s70\n\t"                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"

#define PK_ASM_SLAB_ANY \
        "v_mul_f32_e32 v40, s36, %[ix]\n\t" /* the record holds (corner - origin) already: k_rebase */                 \
        "v_mul_f32_e32 v41, s37, %[iy]\n\t"                                                                            \
        "v_mul_f32_e32 v42, s38, %[iz]\n\t"                                                                            \
        "v_mul_f32_e32 v43, s39, %[ix]\n\t"                                                                            \
        "v_mul_f32_e32 v44, s40, %[iy]\n\t"                                                                            \
        "v_mul_f32_e32 v45, s41, %[iz]\n\t"                                                                            \
        "v_min_f32_e32 v46, v40, v43\n\t"                                                                              \
        "v_min_f32_e32 v47, v41, v44\n\t"                                                                              \
        "v_min_f32_e32 v48, v42, v45\n\t"                                                                              \
        "v_max_f32_e32 v40, v40, v43\n\t"                                                                              \
        "v_max_f32_e32 v41, v41, v44\n\t"                                                                              \
        "v_max_f32_e32 v42, v42, v45\n\t"                                                                              \
        "v_max3_f32 v51, v46, v47, v48\n\t" /* tNear of child 1 */                                                     \
        "v_min3_f32 v50, v40, v41, v42\n\t" /* tFar */                                                                 \
        "v_mul_f32_e32 v40, s42, %[ix]\n\t"                                                                            \
        "v_mul_f32_e32 v41, s43, %[iy]\n\t"                                                                            \
        "v_mul_f32_e32 v42, s44, %[iz]\n\t"                                                                            \
        "v_mul_f32_e32 v43, s45, %[ix]\n\t"                                                                            \
        "v_mul_f32_e32 v44, s46, %[iy]\n\t"                                                                            \
        "v_mul_f32_e32 v45, s47, %[iz]\n\t"                                                                            \
        "v_min_f32_e32 v46, v40, v43\n\t"                                                                              \
        "v_min_f32_e32 v47, v41, v44\n\t"                                                                              \
        "v_min_f32_e32 v48, v42, v45\n\t"                                                                              \
        "v_max_f32_e32 v40, v40, v43\n\t"                                                                              \
        "v_max_f32_e32 v41, v41, v44\n\t"                                                                              \
        "v_max_f32_e32 v42, v42, v45\n\t"                                                                              \
        "v_max3_f32 v52, v46, v47, v48\n\t" /* tNear of child 2 */                                                     \
        "v_min3_f32 v54, v40, v41, v42\n\t" /* tFar */

// The slab tests for a packet whose rays all point the same way along every axis (and have finite reciprocals), on a tree
// whose boxes are finite and not inverted (min <= max): with 1/d > 0 the product with the box's lower corner is the smaller of the two (rounding is
// monotonic, and lo - o <= hi - o), with 1/d < 0 the one with the upper corner, so the min and max of raytrace.glsl:117-118
// are known in advance and 12 of the 28 vector instructions go away.  Same values as the general form (a zero may differ
// in sign where both products are zeros, which no comparison tells apart).
// n1x = the register holding child 1's NEAR plane along x for this direction, f1x its far plane, and so on.
#define PK_ASM_SLAB_SIGNED(n1x, f1x, n2x, f2x, n1y, f1y, n2y, f2y, n1z, f1z, n2z, f2z) \
        "v_mul_f32_e32 v40, " n1x ", %[ix]\n\t"                                                                        \
        "v_mul_f32_e32 v41, " n1y ", %[iy]\n\t"                                                                        \
        "v_mul_f32_e32 v42, " n1z ", %[iz]\n\t"                                                                        \
        "v_mul_f32_e32 v43, " f1x ", %[ix]\n\t"                                                                        \
        "v_mul_f32_e32 v44, " f1y ", %[iy]\n\t"                                                                        \
        "v_mul_f32_e32 v45, " f1z ", %[iz]\n\t"                                                                        \
        "v_max3_f32 v51, v40, v41, v42\n\t" /* tNear of child 1 */                                                     \
        "v_min3_f32 v50, v43, v44, v45\n\t" /* tFar */                                                                 \
        "v_mul_f32_e32 v46, " n2x ", %[ix]\n\t"                                                                        \
        "v_mul_f32_e32 v47, " n2y ", %[iy]\n\t"                                                                        \
        "v_mul_f32_e32 v48, " n2z ", %[iz]\n\t"                                                                        \
        "v_mul_f32_e32 v40, " f2x ", %[ix]\n\t"                                                                        \
        "v_mul_f32_e32 v41, " f2y ", %[iy]\n\t"                                                                        \
        "v_mul_f32_e32 v42, " f2z ", %[iz]\n\t"                                                                        \
        "v_max3_f32 v52, v46, v47, v48\n\t" /* tNear of child 2 */                                                     \
        "v_min3_f32 v54, v40, v41, v42\n\t" /* tFar */
#define PK_ASM_SLAB_EXPAND(...) PK_ASM_SLAB_SIGNED(__VA_ARGS__)
#define PK_XP "s36", "s39", "s42", "s45"   /* rays go up the x axis: near planes are the boxes' lower x, far planes the upper */
#define PK_XN "s39", "s36", "s45", "s42"
#define PK_YP "s37", "s40", "s43", "s46"
#define PK_YN "s40", "s37", "s46", "s43"
#define PK_ZP "s38", "s41", "s44", "s47"
#define PK_ZN "s41", "s38", "s47", "s44"

// One order of descent, written out for each of the two (so that nothing has to be selected by the order at run time):
// ref1 / lanes1 / dist1 = the child the wave enters first, the lanes that want it and their tNear; ref2 / s[l2_lo:l2_hi] / dist2
// the other child; s[dis_lo:dis_hi] = the lanes that want both but in the other order.  They leave the wave here and come back
// for the first child after the second (pushed below it), which keeps every lane's own visit order.
#define PK_ASM_DESCEND(ref1, ref2, lanes1, l2_lo, l2_hi, dist1, dist2, dis_lo, dis_hi, tag) \
        "s_andn2_b64 s[64:65], " lanes1 ", s[" dis_lo ":" dis_hi "]\n\t" /* the lanes that enter the first child now */   \
        "s_cbranch_scc0 Lswap" tag "%=\n\t"                                                                             \
        "s_cmp_gt_u32 s69, %[room]\n\t"                                                                                 \
        "s_cbranch_scc1 Lgiveup%=\n\t"                                                                                  \
        "s_cmp_eq_u64 s[" dis_lo ":" dis_hi "], 0\n\t"                                                                  \
        "s_cbranch_scc1 Lnodis" tag "%=\n\t"                                                                            \
        "s_mov_b32 m0, s69\n\t"                                                                                         \
        "s_nop 0\n\t"                                                                                                   \
        "v_writelane_b32 v57, " ref1 ", m0\n\t"                                                                         \
        "v_writelane_b32 v58, s" dis_lo ", m0\n\t"                                                                      \
        "v_writelane_b32 v59, s" dis_hi ", m0\n\t"                                                                      \
        "v_lshl_add_u32 v55, s69, 8, %[lrow]\n\t"                                                                       \
        "ds_write_b32 v55, " dist1 "\n\t"                                                                               \
        "s_add_i32 s69, s69, 1\n"                                                                                        \
        "Lnodis" tag "%=:\n\t"                                                                                          \
        "s_cmp_eq_u64 s[" l2_lo ":" l2_hi "], 0\n\t"                                                                    \
        "s_cbranch_scc1 Lgo" tag "%=\n\t"                                                                               \
        "s_mov_b32 m0, s69\n\t"                                                                                         \
        "s_nop 0\n\t"                                                                                                   \
        "v_writelane_b32 v57, " ref2 ", m0\n\t"                                                                         \
        "v_writelane_b32 v58, s" l2_lo ", m0\n\t"                                                                       \
        "v_writelane_b32 v59, s" l2_hi ", m0\n\t"                                                                       \
        "v_lshl_add_u32 v55, s69, 8, %[lrow]\n\t"                                                                       \
        "ds_write_b32 v55, " dist2 "\n\t"                                                                               \
        "s_add_i32 s69, s69, 1\n"                                                                                        \
        "Lgo" tag "%=:\n\t"                                                                                             \
        "s_mov_b32 s68, " ref1 "\n\t"                                                                                   \
        "s_mov_b64 exec, s[64:65]\n\t"                                                                                  \
        "s_branch Lnode%=\n"                                                                                             \
        "Lswap" tag "%=:\n\t" /* nobody enters the first child (then nobody dissents): on to the second */              \
        "s_cmp_eq_u64 s[" l2_lo ":" l2_hi "], 0\n\t"                                                                    \
        "s_cbranch_scc1 Lpop%=\n\t"                                                                                     \
        "s_mov_b32 s68, " ref2 "\n\t"                                                                                   \
        "s_mov_b64 exec, s[" l2_lo ":" l2_hi "]\n\t"                                                                    \
        "s_branch Lnode%=\n"

#define PK_ASM_TAIL \
        /* Who wants which child, and in which order (raytrace.glsl:113-125, 173-190).  The reference's distance to a  \
           box is d = met ? (tNear > EPS ? tNear : 0) : inf with met = tFar >= tNear && tFar > EPS, and a lane wants a     \
           child when d < best.  best is +inf or the t of an accepted triangle, which exceeds EPS, so for a box that is    \
           met d < best <=> tNear < best; d1 < d2 between two met boxes <=> tNear1 < tNear2 && tNear2 > EPS; and the pop   \
           test `d < best` of a stacked child reads the same with tNear in place of d.  The masks are therefore built      \
           from tNear and tFar as they are, and d itself is never formed. */                                               \
        "v_cmp_ge_f32_e32 vcc, v50, v51\n\t" /* tFar1 >= tNear1 */                                                       \
        "v_cmp_lt_f32_e64 s[66:67], v53, v50\n\t" /* EPS < tFar1 */                                                      \
        "v_cmp_lt_f32_e64 s[52:53], v51, %[best]\n\t"                                                                    \
        "s_and_b64 s[66:67], vcc, s[66:67]\n\t"                                                                          \
        "s_and_b64 s[52:53], s[52:53], s[66:67]\n\t" /* W1 */                                                            \
        "v_cmp_ge_f32_e32 vcc, v54, v52\n\t" /* tFar2 >= tNear2 */                                                       \
        "v_cmp_lt_f32_e64 s[66:67], v53, v54\n\t" /* EPS < tFar2 */                                                      \
        "v_cmp_lt_f32_e64 s[54:55], v52, %[best]\n\t"                                                                    \
        "s_and_b64 s[66:67], vcc, s[66:67]\n\t"                                                                          \
        "s_and_b64 s[54:55], s[54:55], s[66:67]\n\t" /* W2 */                                                            \
        "v_cmp_lt_f32_e32 vcc, v51, v52\n\t"                                                                             \
        "v_cmp_lt_f32_e64 s[56:57], v53, v52\n\t" /* EPS < tNear2 */                                                     \
        "s_and_b64 s[56:57], s[56:57], vcc\n\t" /* LT: child 1 first */                                                  \
        "s_and_b64 s[58:59], s[52:53], s[54:55]\n\t" /* both */                                                        \
        "s_and_b64 s[60:61], s[58:59], s[56:57]\n\t" /* both, child 1 first */                                         \
        "s_andn2_b64 s[62:63], s[58:59], s[56:57]\n\t" /* both, child 2 first */                                       \
        "s_bcnt1_i32_b64 s70, s[60:61]\n\t"                                                                            \
        "s_bcnt1_i32_b64 s71, s[62:63]\n\t"                                                                            \
        "s_cmp_gt_u32 s70, s71\n\t" /* scc: the wave takes child 1 first */                                            \
        "s_cbranch_scc0 Lsecond%=\n\t"                                                                                 \
        PK_ASM_DESCEND("s48", "s49", "s[52:53]", "54", "55", "v51", "v52", "62", "63", "A")                              \
        "Lsecond%=:\n\t"                                                                                               \
        PK_ASM_DESCEND("s49", "s48", "s[54:55]", "52", "53", "v52", "v51", "60", "61", "B")                              \
        /* ---- leaf: the launch's rebased triangle records, the last one of the leaf has flag bit 0 */              \
        "Lleaf%=:\n\t"                                                                                                 \
        "s_cmp_eq_u32 s68, 0x80000001\n\t"                                                                             \
        "s_cbranch_scc1 Lpop%=\n\t"                                                                                    \
        "s_not_b32 s70, s68\n\t"                                                                                       \
        "s_lshl_b32 s70, s70, 6\n"                                                                                     \
        "Ltri%=:\n\t"                                                                                                  \
        "s_load_dwordx16 s[36:51], %[tris], s70\n\t" /* (e1 flags)(e2 id)(o - a, dot(e2, q))(q = cross(o - a, e1), -): k_rebase_tris */ \
        "s_add_u32 s70, s70, 64\n\t"                                                                                   \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_mul_f32_e32 v40, s42, %[dy]\n\t"                                                                            \
        "v_mul_f32_e32 v41, s41, %[dz]\n\t"                                                                            \
        "v_mul_f32_e32 v42, s40, %[dz]\n\t"                                                                            \
        "v_mul_f32_e32 v43, s42, %[dx]\n\t"                                                                            \
        "v_mul_f32_e32 v44, s41, %[dx]\n\t"                                                                            \
        "v_mul_f32_e32 v45, s40, %[dy]\n\t"                                                                            \
        "v_sub_f32_e32 v40, v40, v41\n\t" /* p = cross(d, e2) */                                                       \
        "v_sub_f32_e32 v41, v42, v43\n\t"                                                                              \
        "v_sub_f32_e32 v42, v44, v45\n\t"                                                                              \
        "v_mul_f32_e32 v46, s36, v40\n\t"                                                                              \
        "v_mul_f32_e32 v47, s37, v41\n\t"                                                                              \
        "v_mul_f32_e32 v48, s38, v42\n\t"                                                                              \
        "v_mul_f32_e32 v40, s44, v40\n\t"                                                                              \
        "v_mul_f32_e32 v41, s45, v41\n\t"                                                                              \
        "v_mul_f32_e32 v42, s46, v42\n\t"                                                                              \
        "v_add_f32_e32 v46, v46, v47\n\t"                                                                              \
        "v_add_f32_e32 v40, v40, v41\n\t"                                                                              \
        "v_add_f32_e32 v44, v48, v46\n\t" /* det = dot(e1, p) */                                                       \
        "v_add_f32_e32 v40, v42, v40\n\t" /* dot(o - a, p) */                                                          \
        "v_mul_f32_e32 v42, s48, %[dx]\n\t"                                                                            \
        "v_mul_f32_e32 v46, s49, %[dy]\n\t"                                                                            \
        "v_div_scale_f32 v48, s[56:57], v44, v44, 1.0\n\t" /* inv_det = 1 / det, correctly rounded */                  \
        "v_rcp_f32_e32 v45, v48\n\t"                                                                                   \
        "v_mul_f32_e32 v47, s50, %[dz]\n\t"                                                                            \
        "v_div_scale_f32 v49, vcc, 1.0, v44, 1.0\n\t"                                                                  \
        "v_fma_f32 v43, -v48, v45, 1.0\n\t"                                                                            \
        "v_fmac_f32_e32 v45, v43, v45\n\t"                                                                             \
        "v_mul_f32_e32 v43, v49, v45\n\t"                                                                              \
        "v_add_f32_e32 v42, v42, v46\n\t"                                                                              \
        "v_fma_f32 v46, -v48, v43, v49\n\t"                                                                            \
        "v_fmac_f32_e32 v43, v46, v45\n\t"                                                                             \
        "v_fma_f32 v46, -v48, v43, v49\n\t"                                                                            \
        "v_div_fmas_f32 v43, v46, v45, v43\n\t"                                                                        \
        "v_div_fixup_f32 v43, v43, v44, 1.0\n\t"                                                                       \
        "v_add_f32_e32 v42, v47, v42\n\t" /* dot(d, q) */                                                              \
        "v_mul_f32_e32 v40, v40, v43\n\t" /* u */                                                                      \
        "v_mul_f32_e32 v42, v42, v43\n\t" /* v */                                                                      \
        "v_mul_f32_e32 v41, s47, v43\n\t" /* t = dot(e2, q) / det */                                                   \
        "v_cmp_gt_f32_e32 vcc, 0, v40\n\t"                                                                             \
        "v_cmp_gt_f32_e64 s[58:59], |v40|, v53\n\t"                                                                    \
        "v_add_f32_e32 v43, -1.0, v40\n\t"                                                                             \
        "v_mov_b32_e32 v61, s43\n\t"                                                                                   \
        "v_cmp_lt_f32_e64 s[60:61], |v44|, v53\n\t"                                                                    \
        "v_cmp_lt_f32_e64 s[62:63], 1.0, v40\n\t"                                                                      \
        "v_add_f32_e32 v40, v40, v42\n\t"                                                                              \
        "s_and_b64 s[58:59], vcc, s[58:59]\n\t"                                                                        \
        "v_cmp_gt_f32_e64 s[64:65], |v43|, v53\n\t"                                                                    \
        "v_cmp_gt_f32_e64 s[66:67], 0, v42\n\t"                                                                        \
        "v_cmp_gt_f32_e64 s[52:53], |v42|, v53\n\t"                                                                    \
        "v_cmp_lt_f32_e32 vcc, 1.0, v40\n\t"                                                                           \
        "v_add_f32_e32 v40, -1.0, v40\n\t"                                                                             \
        "s_or_b64 s[60:61], s[60:61], s[58:59]\n\t"                                                                    \
        "s_and_b64 s[62:63], s[62:63], s[64:65]\n\t"                                                                   \
        "s_and_b64 s[66:67], s[66:67], s[52:53]\n\t"                                                                   \
        "v_cmp_gt_f32_e64 s[58:59], |v40|, v53\n\t"                                                                    \
        "s_or_b64 s[62:63], s[60:61], s[62:63]\n\t"                                                                    \
        "s_and_b64 s[60:61], vcc, s[58:59]\n\t"                                                                        \
        "s_or_b64 s[62:63], s[66:67], s[62:63]\n\t"                                                                    \
        "v_cmp_nlt_f32_e64 s[54:55], v53, v41\n\t"                                                                     \
        "s_or_b64 s[62:63], s[60:61], s[62:63]\n\t"                                                                    \
        "s_or_b64 vcc, s[54:55], s[62:63]\n\t"                                                                         \
        "v_cndmask_b32_e32 v40, v41, v60, vcc\n\t" /* rejected ? inf : t */                                            \
        "v_cmp_lt_f32_e32 vcc, v40, %[best]\n\t"                                                                       \
        "s_bitcmp0_b32 s39, 0\n\t" /* scc: not the leaf's last triangle */                                                                                      \
        "s_nop 1\n\t"                                                                                                  \
        "v_cndmask_b32_e32 %[bid], %[bid], v61, vcc\n\t"                                                               \
        "v_cndmask_b32_e32 %[best], %[best], v40, vcc\n\t"                                                             \
        "s_cbranch_scc1 Ltri%=\n"                                                                                      \
        /* ---- next stack entry that still has a lane whose distance to it is below its closest hit (raytrace.glsl:151) */ \
        "Lpop%=:\n\t"                                                                                                  \
        "s_cmp_eq_u32 s69, 0\n\t"                                                                                      \
        "s_cbranch_scc1 Ldone%=\n\t"                                                                                   \
        "s_add_i32 s69, s69, -1\n\t"                                                                                   \
        "s_nop 0\n\t"                                                                                                  \
        "v_readlane_b32 s56, v58, s69\n\t"                                                                             \
        "v_readlane_b32 s57, v59, s69\n\t"                                                                             \
        "s_mov_b64 exec, s[56:57]\n\t"                                                                                 \
        "v_lshl_add_u32 v55, s69, 8, %[lrow]\n\t"                                                                      \
        "ds_read_b32 v56, v55\n\t"                                                                                     \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "v_cmp_lt_f32_e32 vcc, v56, %[best]\n\t"                                                                       \
        "s_and_b64 exec, exec, vcc\n\t"                                                                                \
        "s_cbranch_scc0 Lpop%=\n\t"                                                                                    \
        "v_readlane_b32 s68, v57, s69\n\t"                                                                             \
        "s_branch Lnode%=\n"                                                                                           \
        "Lgiveup%=:\n\t"                                                                                               \
        "s_mov_b32 %[gave], 1\n"                                                                                       \
        "Ldone%=:\n\t"                                                                                                 \
        "s_mov_b64 exec, s[72:73]"

#define PK_DEFINE_TRAVERSE(NAME, SLAB)                                                                                                   \
    __device__ __forceinline__ bool NAME(unsigned long long nodes, unsigned long long tris, int root, unsigned long long in0, rt_vec3 o, \
                                         rt_vec3 d, rt_vec3 inv, uint32_t lrow, float& best, int& best_id) {                             \
        int gave;                                                                                                                        \
        asm volatile(PK_ASM_HEAD PK_ASM_PAD_V PK_ASM_PAD_S SLAB PK_ASM_TAIL                                                              \
            : [gave] "=&s"(gave), [best] "+v"(best), [bid] "+v"(best_id)                                                                                                     \
            : [nodes] "s"(nodes), [tris] "s"(tris), [root] "s"(root), [in0] "s"(in0), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z),                                           \
              [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z), [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [lrow] "v"(lrow),                                              \
              [room] "n"(PK_STACK - 2)                                                                                                                                       \
            : "memory", "vcc", "scc", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",             \
              "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69",               \
              "s70", "s71", "s72", "s73", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",               \
              "v55", "v56", "v57", "v58", "v59", "v60", "v61");                                                                                                  \
        return gave != 0;                                                                                                                \
    }
PK_DEFINE_TRAVERSE(pk_traverse_any, PK_ASM_SLAB_ANY)
PK_DEFINE_TRAVERSE(pk_traverse_ppp, PK_ASM_SLAB_EXPAND(PK_XP, PK_YP, PK_ZP))
PK_DEFINE_TRAVERSE(pk_traverse_npp, PK_ASM_SLAB_EXPAND(PK_XN, PK_YP, PK_ZP))
PK_DEFINE_TRAVERSE(pk_traverse_pnp, PK_ASM_SLAB_EXPAND(PK_XP, PK_YN, PK_ZP))
PK_DEFINE_TRAVERSE(pk_traverse_nnp, PK_ASM_SLAB_EXPAND(PK_XN, PK_YN, PK_ZP))
PK_DEFINE_TRAVERSE(pk_traverse_ppn, PK_ASM_SLAB_EXPAND(PK_XP, PK_YP, PK_ZN))
PK_DEFINE_TRAVERSE(pk_traverse_npn, PK_ASM_SLAB_EXPAND(PK_XN, PK_YP, PK_ZN))
PK_DEFINE_TRAVERSE(pk_traverse_pnn, PK_ASM_SLAB_EXPAND(PK_XP, PK_YN, PK_ZN))
PK_DEFINE_TRAVERSE(pk_traverse_nnn, PK_ASM_SLAB_EXPAND(PK_XN, PK_YN, PK_ZN))

template <bool COUNT>
__global__ __launch_bounds__(PK_BLOCK, PK_BLOCKS_PER_CU) void k_primary(DScene sc, DCamera cam, PathsArgs a) {
    __shared__ float s_dist[PK_BLOCK / 64][PK_STACK][64];
    const int tid = threadIdx.x, lane = tid & 63;
    float(*const dist_rows)[64] = s_dist[tid >> 6];
    const pk_cf4 nodes = (pk_cf4)(uintptr_t)sc.nodes;
    const pk_cf4 tris = (pk_cf4)(uintptr_t)sc.tris;
    const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane < 32 ? 0u : 1u << (lane - 32);
    Counters tot = {0, 0, 0, 0, 0};
    unsigned long long tot_paths = 0;

    // Persistent waves that pull their work: a wave takes `pk_chunk` consecutive items at a time.  Chunk c belongs to
    // counter c mod PK_COUNTERS; workgroup b uses counter b mod PK_COUNTERS, so the waves sharing an L2 (workgroup b runs
    // on XCD b mod 8) work on neighbouring tiles.  A wave's first chunk is the one with its own rank among the waves of its
    // counter -- no atomic, no queue of 8,192 waves in front of a handful of addresses at the start of the launch -- and
    // the counters start at that number of waves (set on the stream ahead of the launch); an agent-scope atomic in memory
    // hands out the rest.  The workgroups of a launch are not always resident together -- next to a path kernel only half
    // of them fit -- and with a fixed share of the items per workgroup the late ones would start a second round with the
    // chip half empty (one short-lived workgroup per four items, so that the hardware does the balancing, measured 2 %
    // (dragon80k) to 8 % (teapot) slower end to end).
    const uint32_t nctr = pk_counters(gridDim.x), cid = blockIdx.x % nctr;
    uint32_t* const ctr = a.pk_ctr + cid * 16u;
    // The loop's own variables are wave-uniform and kept so by construction (the wave's number through readfirstlane): its
    // branches are scalar ones, and the lanes of a wave cannot reach the grab below apart from each other.
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t item = ((blockIdx.x / nctr * (PK_BLOCK / 64u) + wave) * nctr + cid) * a.pk_chunk - 1u, left = a.pk_chunk + 1u;
    // The items of a chunk are consecutive frames of one tile (or the last frames of one and the first of the next): the
    // tile's pixels, its rays and their root-box test are worked out when the tile changes, not per item.  (The rays are
    // traced for every frame all the same: only what is identical by construction is kept.)
    uint32_t tq = 0u, f = 0u, cur_tile = 0xffffffffu;
    bool fresh = true;   // the item's tile and frame come from a division (first item of a chunk), not from the previous item's
    int gx = 0, gy = 0;
    bool valid = false, in_root = false;
    rt_vec3 o = rt_v3(0.0f, 0.0f, 0.0f), d = o, inv = o;
    // every item is taken once; a wave that has gone round more often than there are items has lost its way and stops
    for (uint32_t rounds = 0; rounds <= a.total_tiles; rounds++) {
        if (left == 1u) {
            uint32_t g = 0;
            if (lane == 0) g = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
            item = (g * nctr + cid) * a.pk_chunk;
            left = a.pk_chunk + 1u;
            fresh = true;
        } else {
            item++;
        }
        left--;
        if (item >= a.total_tiles) break;   // the chunks after this one lie further out still
        if (fresh) {
            tq = item / a.num_frames;
            f = item - tq * a.num_frames;
            fresh = false;
        } else if (++f == a.num_frames) {
            f = 0u;
            tq++;
        }
        if (tq != cur_tile) {
            cur_tile = tq;
            tile_pixel(a, tq, lane, gx, gy);
            valid = gx < cam.width && gy < cam.height;
            primary_ray(cam, gx, gy, o, d, inv);
            const float d0 = box_dist(o, inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2]);
            in_root = valid && d0 != RT_INF;
        }
        float best = RT_INF;
        int best_id = -1;
        Counters cn = {0, 0, 0, 0, 0};
        bool in = in_root;   // does this lane take part in the current node?
        if (COUNT && valid) {
            cn.rays++;
            cn.box += in ? 2u : 1u;   // the reference evaluates the root box a second time to seed its stack
        }
        int cur = __builtin_amdgcn_ballot_w64(in) != 0ull ? sc.root_ref : RT_REF_IDLE;
        if (COUNT && in && cur < 0) cn.pops++;
        int sp = 0;
        int stk_ref = 0, stk_lo = 0, stk_hi = 0;   // lane i of these registers = entry i of the packet's stack
#define PK_PUSH(ref_, mask_)                                                     \
    do {                                                                         \
        const bool top_ = lane == sp;                                            \
        stk_ref = top_ ? (ref_) : stk_ref;                                       \
        stk_lo = top_ ? (int)(uint32_t)(mask_) : stk_lo;                         \
        stk_hi = top_ ? (int)(uint32_t)((mask_) >> 32) : stk_hi;                 \
    } while (0)
        bool gave_up = false;

        if (!COUNT) {   // the hand-written loop; the counting build keeps the C++ one below
            const uint32_t lrow = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&dist_rows[0][lane];
            const unsigned long long in0 = __builtin_amdgcn_ballot_w64(in);
            // the form of the slab tests: one of the eight sign-specific ones when every ray of the packet points the same
            // way along each axis with a finite reciprocal (all tiles but the ones on the image's two centre lines)
            int form = 8;
            if (a.finite_boxes) {
                const unsigned long long px = __builtin_amdgcn_ballot_w64(inv.x > 0.0f && inv.x < RT_INF), nx = __builtin_amdgcn_ballot_w64(inv.x < 0.0f && inv.x > -RT_INF);
                const unsigned long long py = __builtin_amdgcn_ballot_w64(inv.y > 0.0f && inv.y < RT_INF), ny = __builtin_amdgcn_ballot_w64(inv.y < 0.0f && inv.y > -RT_INF);
                const unsigned long long pz = __builtin_amdgcn_ballot_w64(inv.z > 0.0f && inv.z < RT_INF), nz = __builtin_amdgcn_ballot_w64(inv.z < 0.0f && inv.z > -RT_INF);
                const bool ux = (px & in0) == in0 || (nx & in0) == in0, uy = (py & in0) == in0 || (ny & in0) == in0, uz = (pz & in0) == in0 || (nz & in0) == in0;
                if (ux && uy && uz) form = ((nx & in0) == in0 ? 1 : 0) | ((ny & in0) == in0 ? 2 : 0) | ((nz & in0) == in0 ? 4 : 0);
            }
#define PK_GO(F_) gave_up = F_((unsigned long long)(uintptr_t)a.rnodes, (unsigned long long)(uintptr_t)a.rtris, sc.root_ref, in0, o, d, inv, lrow, best, best_id)
            switch (form) {
                case 0: PK_GO(pk_traverse_ppp); break;
                case 1: PK_GO(pk_traverse_npp); break;
                case 2: PK_GO(pk_traverse_pnp); break;
                case 3: PK_GO(pk_traverse_nnp); break;
                case 4: PK_GO(pk_traverse_ppn); break;
                case 5: PK_GO(pk_traverse_npn); break;
                case 6: PK_GO(pk_traverse_pnn); break;
                case 7: PK_GO(pk_traverse_nnn); break;
                default: PK_GO(pk_traverse_any); break;
            }
#undef PK_GO
            cur = RT_REF_IDLE;
        }
#ifdef PK_DEBUG
        uint32_t steps = 0;
#endif
        while (cur != RT_REF_IDLE) {
#ifdef PK_DEBUG
            if (++steps > 100000u) {
                if (lane == 0) printf("k_primary: item %u stuck at node %d sp %d\n", item, cur, sp);
                gave_up = true;
                break;
            }
#endif
            if (cur >= 0) {
                const pk_f4 q0 = nodes[(uint32_t)cur * 4u], q1 = nodes[(uint32_t)cur * 4u + 1u], q2 = nodes[(uint32_t)cur * 4u + 2u],
                            q3 = nodes[(uint32_t)cur * 4u + 3u];
                const int r1 = (int)rt_f2u(q3.x), r2 = (int)rt_f2u(q3.y);
                const float d1 = box_dist(o, inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                const float d2 = box_dist(o, inv, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                if (COUNT && in) {
                    cn.pops++;
                    cn.box += 2;
                }
                const bool w1 = in && d1 < best, w2 = in && d2 < best, lt = d1 < d2;
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(w1), m2 = __builtin_amdgcn_ballot_w64(w2);
                const unsigned long long mlt = __builtin_amdgcn_ballot_w64(lt);
                const unsigned long long both = m1 & m2;
                // the child most of the lanes that want both would visit first (ties: child 2, as a single ray does)
                const int n1 = __builtin_popcountll(both & mlt), n2 = __builtin_popcountll(both & ~mlt);
                const bool first1 = n1 > n2;
                const unsigned long long dissent = both & (first1 ? ~mlt : mlt);
                unsigned long long mf = (first1 ? m1 : m2) & ~dissent, ms = first1 ? m2 : m1;
                int rf = first1 ? r1 : r2, rs = first1 ? r2 : r1;
                float df = first1 ? d1 : d2, ds = first1 ? d2 : d1;
                if (mf == 0ull) {   // nobody enters the first child now (then nobody dissents either): straight on to the second
                    mf = ms, rf = rs, df = ds;
                    ms = 0ull;
                }
                if ((dissent != 0ull ? 1 : 0) + (ms != 0ull ? 1 : 0) + sp > PK_STACK) {
                    gave_up = true;
                    break;
                }
                if (dissent != 0ull) {   // they come back for the first child after the second
                    PK_PUSH(rf, dissent);
                    dist_rows[sp][lane] = df;
                    sp++;
                }
                if (ms != 0ull) {
                    PK_PUSH(rs, ms);
                    dist_rows[sp][lane] = ds;
                    sp++;
                }
                if (mf != 0ull) {
                    cur = rf;
                    in = (((uint32_t)mf & bit_lo) | ((uint32_t)(mf >> 32) & bit_hi)) != 0u;
                    if (COUNT && in && cur < 0) cn.pops++;
                    continue;
                }
            } else if (cur != RT_REF_EMPTY) {
                uint32_t t = (uint32_t)~cur;
                for (;;) {
#ifdef PK_DEBUG
                    if (++steps > 100000u) {
                        if (lane == 0) printf("k_primary: item %u stuck in leaf %u\n", item, t);
                        break;
                    }
#endif
                    const pk_f4 t0 = tris[t * 4u], t1 = tris[t * 4u + 1u], t2 = tris[t * 4u + 2u];
                    const float dist = tri_dist_flat(o, d, rt_v3(t0.x, t0.y, t0.z), rt_v3(t1.x, t1.y, t1.z), rt_v3(t2.x, t2.y, t2.z));
                    if (COUNT && in) cn.tri++;
                    if (in && dist < best) {
                        best = dist;
                        best_id = (int)rt_f2u(t1.w);
                    }
                    if (rt_f2u(t0.w) & 1u) break;
                    t++;
                }
            }
            // next entry of the stack that still has a lane whose distance to it is below its closest hit (raytrace.glsl:151)
            cur = RT_REF_IDLE;
            while (sp > 0) {
                sp--;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(stk_lo, sp), hi = (uint32_t)__builtin_amdgcn_readlane(stk_hi, sp);
                const float dd = dist_rows[sp][lane];
                in = ((lo & bit_lo) | (hi & bit_hi)) != 0u && dd < best;
                if (__builtin_amdgcn_ballot_w64(in) != 0ull) {
                    cur = __builtin_amdgcn_readlane(stk_ref, sp);
                    if (COUNT && in && cur < 0) cn.pops++;
                    break;
                }
            }
        }

#undef PK_PUSH
        // Every lane takes the same way from here to the next item: all lanes store the item's count (same address, same
        // value) rather than one of them under a branch, so that no lane reaches the next grab ahead of the others.
        const bool hit = !gave_up && valid && best < RT_INF;
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
        // a packet that gave up stores nothing and counts nothing: the path kernel traces the tile's rays lane by lane
        if (hit) a.hitq[(size_t)item * 64u + rank] = make_float4(best, rt_u2f((uint32_t)best_id), rt_u2f((uint32_t)lane), 0.0f);
        if (!gave_up && valid && !hit)
            a.samples[(size_t)f * a.pixels + ((uint32_t)gy * (uint32_t)cam.width + (uint32_t)gx)] = make_float4(1.0f, 1.0f, 1.0f, d.y);
        a.hcount[item] = gave_up ? PK_TRACE_ITEM : (uint32_t)__builtin_popcountll(hm);
        if (COUNT && !gave_up) {
            tot.box += cn.box, tot.tri += cn.tri, tot.rays += cn.rays, tot.pops += cn.pops;
            tot_paths += valid ? 1ull : 0ull;
        }
    }
    if (COUNT && a.counters) flush_counters6(tot, tot_paths, a.counters);
}

#endif  // RT_PACKETS_H
