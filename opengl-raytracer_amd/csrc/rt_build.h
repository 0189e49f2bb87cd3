// GPU BVH builder: level-synchronous, top-down, binned SAH (16 bins x 3 axes), one pass over the triangles per
// kernel and level.  It is a NEW algorithm, not the reference's ternary-search builder (bvh.cpp:54-134), so it
// yields a different tree and triangle order; what it keeps from the reference is the cost model
// (bvh.cpp:24-56: SA = 6(dx*dy + dx*dz + dy*dz), split cost 1 + 50 nL SA_L + 50 nR SA_R against 50 n SA for a
// leaf), the depth cap MAX_BVH_DEPTH = 32 and the node semantics the traversal expects (leaf = inclusive triangle
// range in leaf order).  The result can be downloaded in the reference's buffer layout (rt_get_built_scene) so
// that the CPU oracle renders the very same tree.
//
// Per level, over the triangle array in its current order (every node owns a contiguous segment):
//   k_bin        each triangle of an active node adds its box to the node's (axis, bin) accumulators -- count and
//                bounds, as order-independent integer atomics (min/max on an order-preserving key), privatised in
//                LDS when a whole workgroup sits in one node (the top levels);
//   k_split      one lane per active node sweeps the 3 x 15 candidate planes, decides split or leaf, allocates the
//                two children (breadth-first ids) and the node's record id;
//   k_partition  each triangle of a split node moves to its child's segment; destinations come from per-child
//                cursors, aggregated per wave (ballot + mbcnt, one atomic per wave and child).
// Once a node has at most BS_MAX triangles (and lies below the levels that the render kernels stage in LDS) it
// leaves the level loop: k_build_small finishes its whole subtree inside one wavefront, in LDS, with the very same
// binning, cost and partition arithmetic, so the tree is the one the level loop alone would have produced.
// The bins, bounds and split decisions are deterministic; the order of triangles INSIDE a leaf depends on the
// atomics' arrival order in the level loop.
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include <type_traits>

#include "rt_device.h"

#define BB_BINS 16
#define BB_WORDS 7                               // count + 6 bounds keys per bin
#define BB_NODE_WORDS (3 * BB_BINS * BB_WORDS)   // 336 words of accumulators per active node
#define BB_MAX_DEPTH 32                          // MAX_BVH_DEPTH, constants.h:26
#define BB_BLOCK 256

// order-preserving map float -> uint32 so that integer atomicMin/Max act as float min/max
__device__ __forceinline__ uint32_t bb_key(float f) {
    uint32_t u = rt_f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bb_unkey(uint32_t k) { return rt_u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
#define BB_KEY_POS_INF 0xff800000u   // bb_key(+inf)
#define BB_KEY_NEG_INF 0x007fffffu   // bb_key(-inf)

struct BuildState {
    // per triangle (original face index): centroid and box
    float4* tcen;
    float4* tmin;
    float4* tmax;
    // per position: triangle at that position and the node owning it (ping-pong)
    uint32_t* order[2];
    int32_t* node_of[2];
    // per node
    float4* nmin;        // bounds
    float4* nmax;
    float4* cmin;        // centroid bounds
    float4* cmax;
    int32_t* nstart;
    int32_t* ncount;
    int32_t* nchild;     // first child (second = +1) or -1 for a leaf
    int32_t* ndepth;
    int32_t* nrec;       // inner-node record id or -1
    int32_t* nslot;      // slot in this level's active list or -1
    int32_t* nsplit;     // axis | bin << 2 of the split chosen this level, or -1
    // per level
    int32_t* active[2];  // active node lists (ping-pong)
    uint32_t* bins;      // active_count x BB_NODE_WORDS
    int32_t* cursors;    // per node: next free position of its segment during partition
    int32_t* counters;   // [0] nodes allocated, [1] records allocated, [2] next level's active count, [3] max depth,
                         // [4] small subtree roots
    int32_t* nparent;    // parent of every node (-1 for the root), filled after the build for refits
    int32_t* nflag;      // refit: arrivals at an inner node
    int32_t* small;      // roots handed to k_build_small
    float4* scratch;     // 4 float4 per triangle position: node entries of a small subtree before they get ids
};

__device__ __forceinline__ int bb_bin(float c, float lo, float hi) {
    float ext = hi - lo;
    if (!(ext > 0.0f)) return 0;
    int b = (int)((c - lo) * ((float)BB_BINS / ext));
    return b < 0 ? 0 : (b > BB_BINS - 1 ? BB_BINS - 1 : b);
}

// centroid and box per triangle; root bounds by wave reduction + one atomic per wave
__global__ __launch_bounds__(BB_BLOCK) void k_build_prep(const float* verts, const uint32_t* faces, int nf, BuildState s,
                                                         uint32_t* root_keys /*12: nmin nmax cmin cmax*/) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    // lanes past the end carry the neutral elements of the reductions below
    rt_vec3 lo = rt_v3(RT_INF, RT_INF, RT_INF), hi = rt_v3(-RT_INF, -RT_INF, -RT_INF), cel = lo, ceh = hi;
    if (i < nf) {
    uint32_t ia = faces[3 * i], ib = faces[3 * i + 1], ic = faces[3 * i + 2];
    rt_vec3 a = rt_v3(verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]);
    rt_vec3 b = rt_v3(verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]);
    rt_vec3 c = rt_v3(verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]);
    lo = rt_v3(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)));
    hi = rt_v3(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)));
    const rt_vec3 ce = rt_v3(((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f);  // bvh.cpp:156
    cel = ceh = ce;
    s.tcen[i] = make_float4(ce.x, ce.y, ce.z, 0.0f);
    s.tmin[i] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    s.tmax[i] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    s.order[0][i] = (uint32_t)i;
    s.node_of[0][i] = 0;
    }
    // root bounds: reduce over the wave with shuffles, then one atomic per wave and component
    const float v[12] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, cel.x, cel.y, cel.z, ceh.x, ceh.y, ceh.z};
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const bool is_min = k < 3 || (k >= 6 && k < 9);
        float x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float y = __shfl_xor(x, off, 64);
            x = is_min ? fminf(x, y) : fmaxf(x, y);
        }
        // bounds only ever tighten, so a (possibly stale) read that already covers x makes the atomic unnecessary;
        // without the test all waves serialise on twelve addresses
        if ((threadIdx.x & 63) == 0) {
            const uint32_t key = bb_key(x), seen = __atomic_load_n(&root_keys[k], __ATOMIC_RELAXED);
            if (is_min) {
                if (key < seen) atomicMin(&root_keys[k], key);
            } else {
                if (key > seen) atomicMax(&root_keys[k], key);
            }
        }
    }
}

__global__ void k_build_root(BuildState s, const uint32_t* root_keys, int nf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    s.nmin[0] = make_float4(bb_unkey(root_keys[0]), bb_unkey(root_keys[1]), bb_unkey(root_keys[2]), 0.0f);
    s.nmax[0] = make_float4(bb_unkey(root_keys[3]), bb_unkey(root_keys[4]), bb_unkey(root_keys[5]), 0.0f);
    s.cmin[0] = make_float4(bb_unkey(root_keys[6]), bb_unkey(root_keys[7]), bb_unkey(root_keys[8]), 0.0f);
    s.cmax[0] = make_float4(bb_unkey(root_keys[9]), bb_unkey(root_keys[10]), bb_unkey(root_keys[11]), 0.0f);
    s.nstart[0] = 0;
    s.ncount[0] = nf;
    s.nchild[0] = -1;
    s.ndepth[0] = 0;
    s.nrec[0] = -1;
    s.nsplit[0] = -1;
    bool act = nf >= 2;
    s.nslot[0] = act ? 0 : -1;
    s.active[0][0] = 0;
    s.counters[0] = 1;
    s.counters[1] = 0;
    s.counters[2] = act ? 1 : 0;
    s.counters[3] = 0;
    s.counters[4] = 0;
}

__global__ __launch_bounds__(BB_BLOCK) void k_build_clear_bins(uint32_t* bins, size_t words) {
    size_t i = (size_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= words) return;
    uint32_t w = (uint32_t)(i % BB_WORDS);
    bins[i] = w == 0 ? 0u : (w <= 3 ? BB_KEY_POS_INF : BB_KEY_NEG_INF);
}

__device__ __forceinline__ void bb_accumulate(uint32_t* b, float4 lo, float4 hi) {
    atomicAdd(&b[0], 1u);
    atomicMin(&b[1], bb_key(lo.x));
    atomicMin(&b[2], bb_key(lo.y));
    atomicMin(&b[3], bb_key(lo.z));
    atomicMax(&b[4], bb_key(hi.x));
    atomicMax(&b[5], bb_key(hi.y));
    atomicMax(&b[6], bb_key(hi.z));
}

// Wave-wide unsigned min / max with DPP; all 64 lanes must be active; the result is valid in lane 63.
#define BB_DPP_STEP(op, x, ctrl, rmask) x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)(x), (int)(x), ctrl, rmask, 0xf, false))
__device__ __forceinline__ uint32_t bb_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t bb_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t bb_wave_min(uint32_t x) {
    BB_DPP_STEP(bb_umin, x, 0xB1, 0xf);   // quad_perm [1,0,3,2]
    BB_DPP_STEP(bb_umin, x, 0x4E, 0xf);   // quad_perm [2,3,0,1]
    BB_DPP_STEP(bb_umin, x, 0x141, 0xf);  // row_half_mirror
    BB_DPP_STEP(bb_umin, x, 0x140, 0xf);  // row_mirror: every lane holds its row's result
    BB_DPP_STEP(bb_umin, x, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    BB_DPP_STEP(bb_umin, x, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t bb_wave_max(uint32_t x) {
    BB_DPP_STEP(bb_umax, x, 0xB1, 0xf);
    BB_DPP_STEP(bb_umax, x, 0x4E, 0xf);
    BB_DPP_STEP(bb_umax, x, 0x141, 0xf);
    BB_DPP_STEP(bb_umax, x, 0x140, 0xf);
    BB_DPP_STEP(bb_umax, x, 0x142, 0xa);
    BB_DPP_STEP(bb_umax, x, 0x143, 0xc);
    return x;
}

__device__ __forceinline__ void bb_accumulate_n(uint32_t* b, uint32_t n, float4 lo, float4 hi) {
    atomicAdd(&b[0], n);
    atomicMin(&b[1], bb_key(lo.x));
    atomicMin(&b[2], bb_key(lo.y));
    atomicMin(&b[3], bb_key(lo.z));
    atomicMax(&b[4], bb_key(hi.x));
    atomicMax(&b[5], bb_key(hi.y));
    atomicMax(&b[6], bb_key(hi.z));
}

__global__ __launch_bounds__(BB_BLOCK) void k_build_bin(BuildState s, int cur, int n) {
    __shared__ uint32_t l_bins[BB_NODE_WORDS];
    __shared__ int l_node, l_uniform;
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    int node = i < n ? s.node_of[cur][i] : -2;
    int slot = node >= 0 ? s.nslot[node] : -1;
    // is the whole workgroup inside one active node?
    if (threadIdx.x == 0) {
        l_node = node;
        l_uniform = 1;
    }
    __syncthreads();
    if (node != l_node || slot < 0) l_uniform = 0;
    for (int k = threadIdx.x; k < BB_NODE_WORDS; k += BB_BLOCK) {
        uint32_t w = (uint32_t)(k % BB_WORDS);
        l_bins[k] = w == 0 ? 0u : (w <= 3 ? BB_KEY_POS_INF : BB_KEY_NEG_INF);
    }
    __syncthreads();
    const bool uniform = l_uniform != 0;
    if (slot >= 0) {
        uint32_t t = s.order[cur][i];
        float4 ce = s.tcen[t], lo = s.tmin[t], hi = s.tmax[t];
        float4 c0 = s.cmin[node], c1 = s.cmax[node];
        int b[3] = {bb_bin(ce.x, c0.x, c1.x), bb_bin(ce.y, c0.y, c1.y), bb_bin(ce.z, c0.z, c1.z)};
        uint32_t* base = uniform ? l_bins : s.bins + (size_t)slot * BB_NODE_WORDS;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) bb_accumulate(base + (ax * BB_BINS + b[ax]) * BB_WORDS, lo, hi);
    }
    __syncthreads();
    if (uniform) {
        uint32_t* g = s.bins + (size_t)s.nslot[l_node] * BB_NODE_WORDS;
        for (int k = threadIdx.x; k < BB_NODE_WORDS; k += BB_BLOCK) {
            uint32_t w = (uint32_t)(k % BB_WORDS), v = l_bins[k];
            if (w == 0) {
                if (v) atomicAdd(&g[k], v);
            } else if (w <= 3) {
                if (v != BB_KEY_POS_INF) atomicMin(&g[k], v);
            } else {
                if (v != BB_KEY_NEG_INF) atomicMax(&g[k], v);
            }
        }
    }
}

__device__ __forceinline__ float bb_area6(const float* lo, const float* hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 6.0f * ((dx * dy + dx * dz) + dy * dz);
}

// One lane per active node: choose the split, create the children.
__global__ __launch_bounds__(BB_BLOCK) void k_build_split(BuildState s, int cur_list, int num_active, int max_nodes,
                                                          int small_max, int small_min_depth) {
    int a = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (a >= num_active) return;
    const int node = s.active[cur_list][a];
    const uint32_t* bins = s.bins + (size_t)a * BB_NODE_WORDS;
    const int n = s.ncount[node], depth = s.ndepth[node];
    const float4 nlo4 = s.nmin[node], nhi4 = s.nmax[node];
    const float nlo[3] = {nlo4.x, nlo4.y, nlo4.z}, nhi[3] = {nhi4.x, nhi4.y, nhi4.z};
    float best_cost = (50.0f * (float)n) * bb_area6(nlo, nhi);  // leaf cost
    int best_axis = -1, best_bin = 0, best_nl = 0;
    float bl_lo[3], bl_hi[3], br_lo[3], br_hi[3];
    if (depth < BB_MAX_DEPTH) {
        for (int ax = 0; ax < 3; ax++) {
            const uint32_t* bb = bins + ax * BB_BINS * BB_WORDS;
            // suffix unions from the right
            float rlo[BB_BINS][3], rhi[BB_BINS][3];
            int rcnt[BB_BINS];
            float lo[3] = {RT_INF, RT_INF, RT_INF}, hi[3] = {-RT_INF, -RT_INF, -RT_INF};
            int cnt = 0;
            for (int b = BB_BINS - 1; b >= 1; b--) {
                const uint32_t* w = bb + b * BB_WORDS;
                cnt += (int)w[0];
                for (int k = 0; k < 3; k++) {
                    lo[k] = fminf(lo[k], bb_unkey(w[1 + k]));
                    hi[k] = fmaxf(hi[k], bb_unkey(w[4 + k]));
                    rlo[b][k] = lo[k];
                    rhi[b][k] = hi[k];
                }
                rcnt[b] = cnt;
            }
            float llo[3] = {RT_INF, RT_INF, RT_INF}, lhi[3] = {-RT_INF, -RT_INF, -RT_INF};
            int lcnt = 0;
            for (int sp = 0; sp < BB_BINS - 1; sp++) {  // left = bins [0..sp], right = bins [sp+1..15]
                const uint32_t* w = bb + sp * BB_WORDS;
                lcnt += (int)w[0];
                for (int k = 0; k < 3; k++) {
                    llo[k] = fminf(llo[k], bb_unkey(w[1 + k]));
                    lhi[k] = fmaxf(lhi[k], bb_unkey(w[4 + k]));
                }
                int rc = rcnt[sp + 1];
                if (lcnt == 0 || rc == 0) continue;
                float cost = (1.0f + (50.0f * (float)lcnt) * bb_area6(llo, lhi)) + (50.0f * (float)rc) * bb_area6(rlo[sp + 1], rhi[sp + 1]);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = ax;
                    best_bin = sp;
                    best_nl = lcnt;
                    for (int k = 0; k < 3; k++) {
                        bl_lo[k] = llo[k];
                        bl_hi[k] = lhi[k];
                        br_lo[k] = rlo[sp + 1][k];
                        br_hi[k] = rhi[sp + 1][k];
                    }
                }
            }
        }
    }
    s.nslot[node] = -1;
    if (best_axis < 0) {  // leaf
        s.nsplit[node] = -1;
        return;
    }
    int c0 = atomicAdd(&s.counters[0], 2);
    if (c0 + 2 > max_nodes) {  // cannot happen with max_nodes = 2*nf, keep the tree valid anyway
        s.nsplit[node] = -1;
        return;
    }
    s.nrec[node] = atomicAdd(&s.counters[1], 1);
    s.nchild[node] = c0;
    s.nsplit[node] = best_axis | (best_bin << 2);
    atomicMax(&s.counters[3], depth + 1);
    const int start = s.nstart[node];
    const float4 c0lo = s.cmin[node], c0hi = s.cmax[node];
    const float clo[3] = {c0lo.x, c0lo.y, c0lo.z}, chi[3] = {c0hi.x, c0hi.y, c0hi.z};
    // plane between bin best_bin and best_bin+1 on the split axis, for the children's centroid bounds
    float plane = clo[best_axis] + (chi[best_axis] - clo[best_axis]) * ((float)(best_bin + 1) / (float)BB_BINS);
    for (int side = 0; side < 2; side++) {
        int c = c0 + side;
        const float* lo = side ? br_lo : bl_lo;
        const float* hi = side ? br_hi : bl_hi;
        int cnt = side ? n - best_nl : best_nl;
        float ccl[3] = {clo[0], clo[1], clo[2]}, cch[3] = {chi[0], chi[1], chi[2]};
        if (side)
            ccl[best_axis] = plane;
        else
            cch[best_axis] = plane;
        // centroid bounds may not exceed the child's own box
        for (int k = 0; k < 3; k++) {
            ccl[k] = fmaxf(ccl[k], lo[k]);
            cch[k] = fminf(cch[k], hi[k]);
        }
        s.nmin[c] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        s.nmax[c] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        s.cmin[c] = make_float4(ccl[0], ccl[1], ccl[2], 0.0f);
        s.cmax[c] = make_float4(cch[0], cch[1], cch[2], 0.0f);
        s.nstart[c] = side ? start + best_nl : start;
        s.ncount[c] = cnt;
        s.nchild[c] = -1;
        s.ndepth[c] = depth + 1;
        s.nrec[c] = -1;
        s.nsplit[c] = -1;
        s.cursors[c] = side ? start + best_nl : start;
        bool act = cnt >= 2 && depth + 1 < BB_MAX_DEPTH;
        int slot = -1;
        if (act && cnt <= small_max && depth + 1 >= small_min_depth) {
            s.small[atomicAdd(&s.counters[4], 1)] = c;
        } else if (act) {
            slot = atomicAdd(&s.counters[2], 1);
            s.active[cur_list ^ 1][slot] = c;
        }
        s.nslot[c] = slot;
    }
}

// Moves every triangle of a node split this level into its child's segment.
__global__ __launch_bounds__(BB_BLOCK) void k_build_partition(BuildState s, int cur, int n) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    bool in = i < n;
    int node = in ? s.node_of[cur][i] : -1;
    uint32_t t = in ? s.order[cur][i] : 0u;
    int sp = node >= 0 ? s.nsplit[node] : -1;
    int child = -1;
    if (sp >= 0) {
        int ax = sp & 3, bin = sp >> 2;
        float4 ce = s.tcen[t], c0 = s.cmin[node], c1 = s.cmax[node];
        float c = ax == 0 ? ce.x : (ax == 1 ? ce.y : ce.z);
        float lo = ax == 0 ? c0.x : (ax == 1 ? c0.y : c0.z), hi = ax == 0 ? c1.x : (ax == 1 ? c1.y : c1.z);
        child = s.nchild[node] + (bb_bin(c, lo, hi) > bin ? 1 : 0);
    }
    int dest = i;
    // wave-aggregated cursor allocation: one atomic per distinct child present in the wave
    unsigned long long todo = __builtin_amdgcn_ballot_w64(child >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int c = __shfl(child, leader, 64);
        unsigned long long same = __builtin_amdgcn_ballot_w64(child == c);
        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&s.cursors[c], (int)__builtin_popcountll(same));
        base = __shfl(base, leader, 64);
        if (child == c) dest = base + (int)rank;
        todo &= ~same;
    }
    if (in) {
        s.order[cur ^ 1][dest] = t;
        s.node_of[cur ^ 1][dest] = child >= 0 ? child : node;
    }
}

// ---- level loop, second generation: a workgroup owns `span` consecutive positions and walks the node segments in it
// ---------------------------------------------------------------------------------------------------------------------
// Every node owns a contiguous segment of positions, so a workgroup's range is a short sequence of segments.  Each
// segment of an active node is binned into LDS and flushed once: the number of global atomics per level is
// (segments x bins) instead of (triangles x 21), and top levels no longer serialise on a few hot addresses.
__global__ __launch_bounds__(BB_BLOCK) void k_build_bin2(BuildState s, int cur, int n, int span) {
    __shared__ uint32_t l_bins[BB_NODE_WORDS];
    const int b0 = blockIdx.x * span, b1 = min(n, b0 + span);
    int pos = b0;
    while (pos < b1) {
        const int node = s.node_of[cur][pos];
        int end = min(b1, s.nstart[node] + s.ncount[node]);
        if (end <= pos) end = pos + 1;  // cannot happen; guarantees progress
        const int slot = s.nslot[node];
        if (slot >= 0) {
            for (int k = threadIdx.x; k < BB_NODE_WORDS; k += BB_BLOCK) {
                uint32_t w = (uint32_t)(k % BB_WORDS);
                l_bins[k] = w == 0 ? 0u : (w <= 3 ? BB_KEY_POS_INF : BB_KEY_NEG_INF);
            }
            __syncthreads();
            const float4 c0 = s.cmin[node], c1 = s.cmax[node];
            // Neighbouring triangles of a mesh mostly fall into the same bin, which would serialise per-lane LDS atomics
            // 64-fold.  Per axis the wavefront therefore peels off one bin at a time: lanes of the leader's bin are
            // reduced with DPP and one lane issues the seven atomics.  Scattered bins (few lanes each) take the plain
            // per-lane path.
            const int lane = threadIdx.x & 63;
            for (int base = pos; base < end; base += BB_BLOCK) {
                const int i = base + (int)threadIdx.x;
                const bool in = i < end;
                uint32_t klo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, khi[3] = {0u, 0u, 0u};
                int b[3] = {0, 0, 0};
                if (in) {
                    const uint32_t t = s.order[cur][i];
                    const float4 ce = s.tcen[t], lo = s.tmin[t], hi = s.tmax[t];
                    b[0] = bb_bin(ce.x, c0.x, c1.x); b[1] = bb_bin(ce.y, c0.y, c1.y); b[2] = bb_bin(ce.z, c0.z, c1.z);
                    klo[0] = bb_key(lo.x); klo[1] = bb_key(lo.y); klo[2] = bb_key(lo.z);
                    khi[0] = bb_key(hi.x); khi[1] = bb_key(hi.y); khi[2] = bb_key(hi.z);
                }
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    unsigned long long todo = __builtin_amdgcn_ballot_w64(in);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int lb = __builtin_amdgcn_readlane(b[ax], leader);
                        const bool mine = in && b[ax] == lb;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine) & todo;
                        const int members = (int)__builtin_popcountll(mask);
                        if (members < 8) {
                            // scattered: everybody left adds its own triangle
                            if ((todo >> lane) & 1ull) {
                                uint32_t* w = l_bins + (ax * BB_BINS + b[ax]) * BB_WORDS;
                                atomicAdd(&w[0], 1u);
                                atomicMin(&w[1], klo[0]); atomicMin(&w[2], klo[1]); atomicMin(&w[3], klo[2]);
                                atomicMax(&w[4], khi[0]); atomicMax(&w[5], khi[1]); atomicMax(&w[6], khi[2]);
                            }
                            break;
                        }
                        const bool sel = (mask >> lane) & 1ull;
                        const uint32_t m0 = bb_wave_min(sel ? klo[0] : 0xffffffffu), m1 = bb_wave_min(sel ? klo[1] : 0xffffffffu),
                                       m2 = bb_wave_min(sel ? klo[2] : 0xffffffffu);
                        const uint32_t x0 = bb_wave_max(sel ? khi[0] : 0u), x1 = bb_wave_max(sel ? khi[1] : 0u),
                                       x2 = bb_wave_max(sel ? khi[2] : 0u);
                        if (lane == 63) {  // the DPP reductions end in the last lane
                            uint32_t* w = l_bins + (ax * BB_BINS + lb) * BB_WORDS;
                            atomicAdd(&w[0], (uint32_t)members);
                            atomicMin(&w[1], m0); atomicMin(&w[2], m1); atomicMin(&w[3], m2);
                            atomicMax(&w[4], x0); atomicMax(&w[5], x1); atomicMax(&w[6], x2);
                        }
                        todo &= ~mask;
                    }
                }
            }
            __syncthreads();
            uint32_t* g = s.bins + (size_t)slot * BB_NODE_WORDS;
            for (int k = threadIdx.x; k < BB_NODE_WORDS; k += BB_BLOCK) {
                const uint32_t w = (uint32_t)(k % BB_WORDS), v = l_bins[k];
                if (w == 0) {
                    if (v) atomicAdd(&g[k], v);
                } else if (w <= 3) {
                    if (v != BB_KEY_POS_INF) atomicMin(&g[k], v);
                } else {
                    if (v != BB_KEY_NEG_INF) atomicMax(&g[k], v);
                }
            }
            __syncthreads();
        }
        pos = end;
    }
}

#define BB_MAX_SPAN 2048
// Partition with one cursor update per (workgroup, segment): lefts are counted per 64-position chunk, prefix-summed in
// LDS, and the segment's part of the workgroup moves as two stable runs.
__global__ __launch_bounds__(BB_BLOCK) void k_build_partition2(BuildState s, int cur, int n, int span) {
    __shared__ int l_cnt[BB_MAX_SPAN / 64];
    __shared__ int l_base[2];
    const int b0 = blockIdx.x * span, b1 = min(n, b0 + span);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pos = b0;
    while (pos < b1) {
        const int node = s.node_of[cur][pos];
        int end = min(b1, s.nstart[node] + s.ncount[node]);
        if (end <= pos) end = pos + 1;
        const int sp = s.nsplit[node];
        if (sp < 0) {
            for (int i = pos + (int)threadIdx.x; i < end; i += BB_BLOCK) {
                s.order[cur ^ 1][i] = s.order[cur][i];
                s.node_of[cur ^ 1][i] = node;
            }
        } else {
            const int ax = sp & 3, bin = sp >> 2, len = end - pos, child0 = s.nchild[node];
            const float4 c0 = s.cmin[node], c1 = s.cmax[node];
            const float lo = ax == 0 ? c0.x : (ax == 1 ? c0.y : c0.z), hi = ax == 0 ? c1.x : (ax == 1 ? c1.y : c1.z);
            uint32_t rightbits = 0;
            int k = 0;
            for (int cb = wave * 64; cb < len; cb += BB_BLOCK, k++) {
                const int i = pos + cb + lane;
                const bool in = i < end;
                bool right = false;
                if (in) {
                    const float4 ce = s.tcen[s.order[cur][i]];
                    right = bb_bin(ax == 0 ? ce.x : (ax == 1 ? ce.y : ce.z), lo, hi) > bin;
                }
                const unsigned long long bl = __builtin_amdgcn_ballot_w64(in && !right);
                if (lane == 0) l_cnt[cb >> 6] = (int)__builtin_popcountll(bl);
                rightbits |= (right ? 1u : 0u) << k;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const int chunks = (len + 63) >> 6;
                int tot = 0;
                for (int c = 0; c < chunks; c++) {
                    const int v = l_cnt[c];
                    l_cnt[c] = tot;
                    tot += v;
                }
                l_base[0] = tot ? atomicAdd(&s.cursors[child0], tot) : 0;
                l_base[1] = len - tot ? atomicAdd(&s.cursors[child0 + 1], len - tot) : 0;
            }
            __syncthreads();
            k = 0;
            for (int cb = wave * 64; cb < len; cb += BB_BLOCK, k++) {
                const int i = pos + cb + lane;
                const bool in = i < end, right = ((rightbits >> k) & 1u) != 0;
                const unsigned long long bl = __builtin_amdgcn_ballot_w64(in && !right);
                const unsigned long long br = __builtin_amdgcn_ballot_w64(in && right);
                const unsigned long long mine = right ? br : bl;
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
                const int lefts_before = l_cnt[cb >> 6];
                if (in) {
                    const int dest = right ? l_base[1] + (cb - lefts_before) + rank : l_base[0] + lefts_before + rank;
                    s.order[cur ^ 1][dest] = s.order[cur][i];
                    s.node_of[cur ^ 1][dest] = child0 + (right ? 1 : 0);
                }
            }
            __syncthreads();
        }
        pos = end;
    }
}

// ---- small subtrees: one wavefront finishes a node of at most BS_MAX triangles -------------------------------------
#define BS_MAX 256        // most triangles a small subtree may have (the kernel is instantiated for 128 and 256)
#define BS_TINY_MAX 8     // most triangles of a node that a single lane finishes (second phase of the kernel)
#define BS_STACK 36       // pending right siblings, one per level below the root (depth cap 32)
#define BS_MIN_DEPTH 9    // the level loop keeps the levels whose records the render kernels stage in LDS

__device__ __forceinline__ float bs_sel(int ax, float x, float y, float z) { return ax == 0 ? x : (ax == 1 ? y : z); }

__device__ __forceinline__ float bs_bcast(float x, int src_lane) {  // src_lane must be wave-uniform
    return rt_u2f((uint32_t)__builtin_amdgcn_readlane((int)rt_f2u(x), src_lane));
}
// minimum over each row of 16 lanes (all lanes active), left in every lane of the row
__device__ __forceinline__ float bs_row_min(float x) {
#define BS_DPP_MIN(ctrl)                                                                                      \
    x = fminf(x, rt_u2f((uint32_t)__builtin_amdgcn_update_dpp((int)rt_f2u(x), (int)rt_f2u(x), ctrl, 0xf, 0xf, false)))
    BS_DPP_MIN(0xB1);   // quad_perm [1,0,3,2]
    BS_DPP_MIN(0x4E);   // quad_perm [2,3,0,1]
    BS_DPP_MIN(0x141);  // row_half_mirror
    BS_DPP_MIN(0x140);  // row_mirror
#undef BS_DPP_MIN
    return x;
}

// DPP helpers for scans along a row of 16 lanes: lanes whose source lies outside the row receive `ident`
#define BS_DPP(x, ctrl, ident) ((uint32_t)__builtin_amdgcn_update_dpp((int)(ident), (int)(x), ctrl, 0xf, 0xf, false))
#define BS_SCAN_STEP(op, x, ctrl, ident) x = op(x, BS_DPP(x, ctrl, ident))
__device__ __forceinline__ uint32_t bs_uadd(uint32_t a, uint32_t b) { return a + b; }

// Orders the LDS traffic of a one-wavefront workgroup: its DS instructions execute in program order, so a compiler
// fence is all that is needed -- __syncthreads() would also wait for the outstanding global stores (microseconds).
__device__ __forceinline__ void bs_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct SmallNode {  // wave-uniform description of the node being worked on
    float lo[3], hi[3];    // box
    float clo[3], chi[3];  // centroid bounds (the binning interval)
    int id, depth, a, cnt; // local id, depth, first position (relative) and triangle count
};

// tells the compiler that the node description is the same in every lane (it is: it comes from uniform addresses)
__device__ __forceinline__ void bs_make_uniform(SmallNode& n) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        n.lo[k] = rt_u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)rt_f2u(n.lo[k])));
        n.hi[k] = rt_u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)rt_f2u(n.hi[k])));
        n.clo[k] = rt_u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)rt_f2u(n.clo[k])));
        n.chi[k] = rt_u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)rt_f2u(n.chi[k])));
    }
    n.id = __builtin_amdgcn_readfirstlane(n.id);
    n.depth = __builtin_amdgcn_readfirstlane(n.depth);
    n.a = __builtin_amdgcn_readfirstlane(n.a);
    n.cnt = __builtin_amdgcn_readfirstlane(n.cnt);
}

// queue entry of the second phase: centroid bounds, id | depth << 16, first position | count << 16 (the node's box is
// the union of its triangles' boxes and is recomputed by the lane that takes the entry)
__device__ __forceinline__ void bs_queue_put(float* e, const float* clo, const float* chi, int id, int depth, int a, int cnt) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        e[k] = clo[k];
        e[3 + k] = chi[k];
    }
    e[6] = rt_u2f((uint32_t)id | ((uint32_t)depth << 16));
    e[7] = rt_u2f((uint32_t)a | ((uint32_t)cnt << 16));
}

__device__ __forceinline__ void bs_write_entry(float4* sn, int id, const float* lo, const float* hi, int child, int a, int cnt, int rec) {
    sn[2 * id] = make_float4(lo[0], lo[1], lo[2], rt_u2f((uint32_t)child));
    sn[2 * id + 1] = make_float4(hi[0], hi[1], hi[2], rt_u2f((uint32_t)a | ((uint32_t)cnt << 12) | ((uint32_t)rec << 24)));
}

// Two phases.  (1) The wavefront walks the subtree depth-first, one node at a time, 45 lanes = 45 candidate planes.
// Nodes of at most TINY triangles are not visited but queued.  (2) The queue is drained 64 nodes at a time, ONE LANE PER
// NODE: with so few triangles a lane evaluates only the planes that start at one of its triangles' bins (the cost is
// constant between them and the sweep keeps the first plane of a run), splits, and queues the children that need
// another visit.  On a wave64 SIMD a vector instruction costs the same whether one or 64 lanes work, and tiny nodes
// are most of the tree.
template <int SMAX, int TINY>
__global__ __launch_bounds__(64) void k_build_small(BuildState s, int cur, int num_small) {
    // rings of queued nodes, by size class; queued nodes own disjoint sets of triangles (2..4 and 5..8 of them)
    constexpr int Q4CAP = TINY > 0 ? SMAX / 2 : 1, Q8CAP = TINY > 4 ? SMAX / 4 : 1;
    __shared__ float t_ce[3][SMAX], t_lo[3][SMAX], t_hi[3][SMAX];  // triangle data by slot (initial position)
    __shared__ uint32_t t_id[SMAX];                                 // face index by slot
    __shared__ uint16_t l_idx[SMAX];                                // position -> slot
    __shared__ uint16_t l_leaf[SMAX];                               // position -> local id of the node it ends in
    __shared__ uint32_t l_bins[BB_NODE_WORDS];
    __shared__ float l_stk[BS_STACK][16];
    __shared__ float l_q4[Q4CAP][8], l_q8[Q8CAP][8];
    __shared__ int l_cnt[5];                                        // second phase: nodes, records, max depth, queue tails
    __shared__ int l_alloc[2];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= num_small) return;
    const int root = s.small[blockIdx.x];
    const int start = s.nstart[root], n = s.ncount[root];
    if (n > SMAX || n < 2) return;  // cannot happen (k_build_split hands over 2..SMAX triangles)
    float4* sn = s.scratch + (size_t)start * 4;

    for (int p = lane; p < n; p += 64) {
        const uint32_t t = s.order[cur][start + p];
        const float4 ce = s.tcen[t], lo = s.tmin[t], hi = s.tmax[t];
        t_id[p] = t;
        t_ce[0][p] = ce.x; t_ce[1][p] = ce.y; t_ce[2][p] = ce.z;
        t_lo[0][p] = lo.x; t_lo[1][p] = lo.y; t_lo[2][p] = lo.z;
        t_hi[0][p] = hi.x; t_hi[1][p] = hi.y; t_hi[2][p] = hi.z;
        l_idx[p] = (uint16_t)p;
        l_leaf[p] = 0;
    }
    SmallNode nd;
    {
        const float4 lo = s.nmin[root], hi = s.nmax[root], cl = s.cmin[root], ch = s.cmax[root];
        nd.lo[0] = lo.x; nd.lo[1] = lo.y; nd.lo[2] = lo.z;
        nd.hi[0] = hi.x; nd.hi[1] = hi.y; nd.hi[2] = hi.z;
        nd.clo[0] = cl.x; nd.clo[1] = cl.y; nd.clo[2] = cl.z;
        nd.chi[0] = ch.x; nd.chi[1] = ch.y; nd.chi[2] = ch.z;
        nd.id = 0;
        nd.depth = s.ndepth[root];
        nd.a = 0;
        nd.cnt = n;
        bs_make_uniform(nd);
    }
    int m = 1, nrec = 0, maxd = nd.depth, sp = 0, qt4 = 0, qt8 = 0;
    bool have = true;
    if (TINY > 0 && n <= TINY) {  // the whole subtree is one lane's work
        if (n <= 4) {
            if (lane == 0) bs_queue_put(l_q4[0], nd.clo, nd.chi, nd.id, nd.depth, nd.a, nd.cnt);
            qt4 = 1;
        } else {
            if (lane == 0) bs_queue_put(l_q8[0], nd.clo, nd.chi, nd.id, nd.depth, nd.a, nd.cnt);
            qt8 = 1;
        }
        have = false;
    }
    bs_sync();

    while (have) {
        const int cnt = nd.cnt;
        // ---- this lane's triangle of the node's first 64 positions, kept in registers for the whole visit -------------
        const bool in0 = lane < cnt;
        uint32_t slot0 = 0u;
        float q_lo[3] = {RT_INF, RT_INF, RT_INF}, q_hi[3] = {-RT_INF, -RT_INF, -RT_INF};
        int q_b[3] = {0, 0, 0};
        if (in0) {
            slot0 = l_idx[nd.a + lane];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                q_lo[k] = t_lo[k][slot0];
                q_hi[k] = t_hi[k][slot0];
                q_b[k] = bb_bin(t_ce[k][slot0], nd.clo[k], nd.chi[k]);
            }
        }
        // ---- candidate planes: lane = axis * 16 + plane, the lexicographic order of the level loop's sweep ----------
        const int ax = min(lane >> 4, 2), spl = lane & 15;
        const bool cand = lane < 48 && spl < BB_BINS - 1;
        float Llo[3] = {RT_INF, RT_INF, RT_INF}, Lhi[3] = {-RT_INF, -RT_INF, -RT_INF};
        float Rlo[3] = {RT_INF, RT_INF, RT_INF}, Rhi[3] = {-RT_INF, -RT_INF, -RT_INF};
        int lc = 0, rc = 0;
        {
            for (int k = lane; k < BB_NODE_WORDS; k += 64) {
                const uint32_t w = (uint32_t)(k % BB_WORDS);
                l_bins[k] = w == 0 ? 0u : (w <= 3 ? BB_KEY_POS_INF : BB_KEY_NEG_INF);
            }
            bs_sync();
            if (in0) {
                const float4 lo = make_float4(q_lo[0], q_lo[1], q_lo[2], 0.0f), hi = make_float4(q_hi[0], q_hi[1], q_hi[2], 0.0f);
#pragma unroll
                for (int a2 = 0; a2 < 3; a2++) bb_accumulate(l_bins + (a2 * BB_BINS + q_b[a2]) * BB_WORDS, lo, hi);
            }
            for (int p = 64 + lane; p < cnt; p += 64) {
                const uint32_t slot = l_idx[nd.a + p];
                const float4 lo = make_float4(t_lo[0][slot], t_lo[1][slot], t_lo[2][slot], 0.0f);
                const float4 hi = make_float4(t_hi[0][slot], t_hi[1][slot], t_hi[2][slot], 0.0f);
#pragma unroll
                for (int a2 = 0; a2 < 3; a2++)
                    bb_accumulate(l_bins + (a2 * BB_BINS + bb_bin(t_ce[a2][slot], nd.clo[a2], nd.chi[a2])) * BB_WORDS, lo, hi);
            }
            bs_sync();
            // every lane takes ITS bin; the unions left and right of each plane are prefix / suffix scans along the row
            const uint32_t* w = l_bins + (ax * BB_BINS + spl) * BB_WORDS;
            const uint32_t c = w[0];
            uint32_t pc = c, p1 = w[1], p2 = w[2], p3 = w[3], p4 = w[4], p5 = w[5], p6 = w[6];
            uint32_t sc = pc, s1 = p1, s2 = p2, s3 = p3, s4 = p4, s5 = p5, s6 = p6;
#define BS_SCAN(ctrl)                                                                                          \
    BS_SCAN_STEP(bs_uadd, pc, 0x110 + ctrl, 0u);                                                               \
    BS_SCAN_STEP(bb_umin, p1, 0x110 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umin, p2, 0x110 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umin, p3, 0x110 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umax, p4, 0x110 + ctrl, BB_KEY_NEG_INF);                                                   \
    BS_SCAN_STEP(bb_umax, p5, 0x110 + ctrl, BB_KEY_NEG_INF);                                                   \
    BS_SCAN_STEP(bb_umax, p6, 0x110 + ctrl, BB_KEY_NEG_INF);                                                   \
    BS_SCAN_STEP(bs_uadd, sc, 0x100 + ctrl, 0u);                                                               \
    BS_SCAN_STEP(bb_umin, s1, 0x100 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umin, s2, 0x100 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umin, s3, 0x100 + ctrl, BB_KEY_POS_INF);                                                   \
    BS_SCAN_STEP(bb_umax, s4, 0x100 + ctrl, BB_KEY_NEG_INF);                                                   \
    BS_SCAN_STEP(bb_umax, s5, 0x100 + ctrl, BB_KEY_NEG_INF);                                                   \
    BS_SCAN_STEP(bb_umax, s6, 0x100 + ctrl, BB_KEY_NEG_INF)
            BS_SCAN(1);  // row_shr:n feeds the prefixes (bins 0..plane), row_shl:n the suffixes (bins plane..15)
            BS_SCAN(2);
            BS_SCAN(4);
            BS_SCAN(8);
#undef BS_SCAN
            // right side of plane spl = suffix of bin spl + 1
            lc = (int)pc;
            rc = (int)BS_DPP(sc, 0x101, 0u);
            Llo[0] = bb_unkey(p1); Llo[1] = bb_unkey(p2); Llo[2] = bb_unkey(p3);
            Lhi[0] = bb_unkey(p4); Lhi[1] = bb_unkey(p5); Lhi[2] = bb_unkey(p6);
            Rlo[0] = bb_unkey(BS_DPP(s1, 0x101, BB_KEY_POS_INF)); Rlo[1] = bb_unkey(BS_DPP(s2, 0x101, BB_KEY_POS_INF));
            Rlo[2] = bb_unkey(BS_DPP(s3, 0x101, BB_KEY_POS_INF));
            Rhi[0] = bb_unkey(BS_DPP(s4, 0x101, BB_KEY_NEG_INF)); Rhi[1] = bb_unkey(BS_DPP(s5, 0x101, BB_KEY_NEG_INF));
            Rhi[2] = bb_unkey(BS_DPP(s6, 0x101, BB_KEY_NEG_INF));
        }
        // same cost expressions as k_build_split; RT_INF = "not a candidate"
        const float leaf_cost = (50.0f * (float)cnt) * bb_area6(nd.lo, nd.hi);
        float cost = RT_INF;
        if (cand && lc > 0 && rc > 0 && nd.depth < BB_MAX_DEPTH) {
            const float cst = (1.0f + (50.0f * (float)lc) * bb_area6(Llo, Lhi)) + (50.0f * (float)rc) * bb_area6(Rlo, Rhi);
            if (cst < leaf_cost) cost = cst;
        }
        const float rmn = bs_row_min(cost);  // rows 0..2 hold the three axes
        const float mn = fminf(fminf(bs_bcast(rmn, 0), bs_bcast(rmn, 16)), bs_bcast(rmn, 32));
        const unsigned long long win = __builtin_amdgcn_ballot_w64(cost == mn && cost < RT_INF);

        SmallNode next[2];
        bool act[2] = {false, false};
        if (win == 0ull) {
            // leaf: its positions already carry its id
            if (lane == 0) bs_write_entry(sn, nd.id, nd.lo, nd.hi, -1, nd.a, cnt, 0);
        } else {
            const int wl = __ffsll((long long)win) - 1;  // first (axis, plane) of minimum cost, as a strict '<' sweep keeps
            const int best_axis = wl >> 4, best_bin = wl & 15;
            const int nl = __builtin_amdgcn_readlane(lc, wl);
            float bl_lo[3], bl_hi[3], br_lo[3], br_hi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bl_lo[k] = bs_bcast(Llo[k], wl);
                bl_hi[k] = bs_bcast(Lhi[k], wl);
                br_lo[k] = bs_bcast(Rlo[k], wl);
                br_hi[k] = bs_bcast(Rhi[k], wl);
            }
            const int c0 = m, rec = nrec;
            m += 2;
            nrec++;
            maxd = max(maxd, nd.depth + 1);
            if (lane == 0) bs_write_entry(sn, nd.id, nd.lo, nd.hi, c0, nd.a, cnt, rec);
            const float ca = bs_sel(best_axis, nd.clo[0], nd.clo[1], nd.clo[2]), cb2 = bs_sel(best_axis, nd.chi[0], nd.chi[1], nd.chi[2]);
            const float plane = ca + (cb2 - ca) * ((float)(best_bin + 1) / (float)BB_BINS);
#pragma unroll
            for (int side = 0; side < 2; side++) {
                SmallNode& ch = next[side];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    ch.lo[k] = side ? br_lo[k] : bl_lo[k];
                    ch.hi[k] = side ? br_hi[k] : bl_hi[k];
                    float ccl = nd.clo[k], cch = nd.chi[k];
                    if (k == best_axis) {
                        if (side) ccl = plane; else cch = plane;
                    }
                    ch.clo[k] = fmaxf(ccl, ch.lo[k]);  // centroid bounds may not exceed the child's own box
                    ch.chi[k] = fminf(cch, ch.hi[k]);
                }
                ch.id = c0 + side;
                ch.depth = nd.depth + 1;
                ch.a = side ? nd.a + nl : nd.a;
                ch.cnt = side ? cnt - nl : nl;
                act[side] = ch.cnt >= 2 && ch.depth < BB_MAX_DEPTH;
                if (!act[side] && lane == side) bs_write_entry(sn, ch.id, ch.lo, ch.hi, -1, ch.a, ch.cnt, 0);
                if (TINY > 0 && act[side] && ch.cnt <= TINY) {  // left to the second phase
                    if (ch.cnt <= 4) {
                        if (lane == side) bs_queue_put(l_q4[qt4 & (Q4CAP - 1)], ch.clo, ch.chi, ch.id, ch.depth, ch.a, ch.cnt);
                        qt4++;
                    } else {
                        if (lane == side) bs_queue_put(l_q8[qt8 & (Q8CAP - 1)], ch.clo, ch.chi, ch.id, ch.depth, ch.a, ch.cnt);
                        qt8++;
                    }
                    act[side] = false;
                }
            }
            // ---- stable partition of the node's positions (at most BS_MAX / 64 chunks, all read before any write) ----
            uint32_t slot_r[SMAX / 64];
            bool right_r[SMAX / 64];
            slot_r[0] = slot0;
            right_r[0] = in0 && (best_axis == 0 ? q_b[0] : (best_axis == 1 ? q_b[1] : q_b[2])) > best_bin;
            if (cnt > 64) {
#pragma unroll
                for (int k = 1; k < SMAX / 64; k++) {
                    const int p = k * 64 + lane;
                    slot_r[k] = 0;
                    right_r[k] = false;
                    if (p < cnt) {
                        slot_r[k] = l_idx[nd.a + p];
                        right_r[k] = bb_bin(t_ce[best_axis][slot_r[k]], ca, cb2) > best_bin;
                    }
                }
                bs_sync();
            }
            int lbase = 0, rbase = 0;
#pragma unroll
            for (int k = 0; k < SMAX / 64; k++) {
                if (k > 0 && cnt <= 64 * k) break;
                const int p = k * 64 + lane;
                const bool in = p < cnt;
                const unsigned long long bl = __builtin_amdgcn_ballot_w64(in && !right_r[k]);
                const unsigned long long br = __builtin_amdgcn_ballot_w64(in && right_r[k]);
                const unsigned long long mine = right_r[k] ? br : bl;
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
                if (in) {
                    const int dest = nd.a + (right_r[k] ? nl + rbase + rank : lbase + rank);
                    l_idx[dest] = (uint16_t)slot_r[k];
                    l_leaf[dest] = (uint16_t)(c0 + (right_r[k] ? 1 : 0));
                }
                lbase += (int)__builtin_popcountll(bl);
                rbase += (int)__builtin_popcountll(br);
            }
            bs_sync();
        }
        // ---- next node: left child, else right child, else the most recent pending sibling ----------------------------
        if (act[0]) {
            if (act[1] && sp < BS_STACK) {
                if (lane == 0) {
                    float* e = l_stk[sp];
                    for (int k = 0; k < 3; k++) {
                        e[k] = next[1].lo[k]; e[3 + k] = next[1].hi[k]; e[6 + k] = next[1].clo[k]; e[9 + k] = next[1].chi[k];
                    }
                    e[12] = rt_u2f((uint32_t)next[1].id); e[13] = rt_u2f((uint32_t)next[1].depth);
                    e[14] = rt_u2f((uint32_t)next[1].a); e[15] = rt_u2f((uint32_t)next[1].cnt);
                }
                sp++;
            } else if (act[1]) {
                if (lane == 0) bs_write_entry(sn, next[1].id, next[1].lo, next[1].hi, -1, next[1].a, next[1].cnt, 0);  // cannot happen
            }
            nd = next[0];
        } else if (act[1]) {
            nd = next[1];
        } else if (sp > 0) {
            sp--;
            bs_sync();
            const float* e = l_stk[sp];
            for (int k = 0; k < 3; k++) {
                nd.lo[k] = e[k]; nd.hi[k] = e[3 + k]; nd.clo[k] = e[6 + k]; nd.chi[k] = e[9 + k];
            }
            nd.id = (int)rt_f2u(e[12]); nd.depth = (int)rt_f2u(e[13]); nd.a = (int)rt_f2u(e[14]); nd.cnt = (int)rt_f2u(e[15]);
            bs_make_uniform(nd);
            bs_sync();
        } else {
            have = false;
        }
    }

    // ---- second phase: one lane per queued node -----------------------------------------------------------------------------
    if (TINY > 0 && (qt4 > 0 || qt8 > 0)) {
        if (lane == 0) {
            l_cnt[0] = m;
            l_cnt[1] = nrec;
            l_cnt[2] = maxd;
            l_cnt[3] = qt4;
            l_cnt[4] = qt8;
        }
        int head4 = 0, head8 = 0;
        // one batch: up to 64 nodes of at most TN triangles each, popped from `ring`
        auto batch = [&](auto tn_tag, float (*ring)[8], int cap_mask, int& head, int tail) {
            constexpr int TN = decltype(tn_tag)::value;
            const int avail = tail - head;
            const int nb = avail < 64 ? avail : 64;
            const bool on = lane < nb;
            float clo[3] = {0.0f, 0.0f, 0.0f}, chi[3] = {0.0f, 0.0f, 0.0f};
            int id = 0, depth = 0, a = 0, cnt = 0;
            if (on) {
                const float* e = ring[(head + lane) & cap_mask];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    clo[k] = e[k];
                    chi[k] = e[3 + k];
                }
                const uint32_t w0 = rt_f2u(e[6]), w1 = rt_f2u(e[7]);
                id = (int)(w0 & 0xffffu); depth = (int)(w0 >> 16); a = (int)(w1 & 0xffffu); cnt = (int)(w1 >> 16);
            }
            head += nb;
            bs_sync();  // every entry of the batch is in registers before anything is appended
            if (!on) return;
            // the node's triangles (slot, box, packed bins) and its box, the union of theirs
            uint32_t slot[TN];
            float tlo[TN][3], thi[TN][3];
            int tb[TN];
            float nlo[3] = {RT_INF, RT_INF, RT_INF}, nhi[3] = {-RT_INF, -RT_INF, -RT_INF};
#pragma unroll
            for (int i = 0; i < TN; i++) {
                slot[i] = 0u;
                tb[i] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    tlo[i][k] = RT_INF;
                    thi[i][k] = -RT_INF;
                }
                if (i < cnt) {
                    slot[i] = l_idx[a + i];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        tlo[i][k] = t_lo[k][slot[i]];
                        thi[i][k] = t_hi[k][slot[i]];
                        tb[i] |= bb_bin(t_ce[k][slot[i]], clo[k], chi[k]) << (4 * k);
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    nlo[k] = fminf(nlo[k], tlo[i][k]);
                    nhi[k] = fmaxf(nhi[k], thi[i][k]);
                }
            }
            // candidates: per axis the planes that start at a triangle's bin; cost and tie rules of k_build_split
            const float leaf_cost = (50.0f * (float)cnt) * bb_area6(nlo, nhi);
            float bcost = RT_INF;
            int bax = 0, bsp = 0, bnl = 0;
            float BLlo[3], BLhi[3], BRlo[3], BRhi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                BLlo[k] = BRlo[k] = RT_INF;
                BLhi[k] = BRhi[k] = -RT_INF;
            }
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
#pragma unroll
                for (int i = 0; i < TN; i++) {
                    const int spl = (tb[i] >> (4 * ax)) & 15;
                    float Llo[3] = {RT_INF, RT_INF, RT_INF}, Lhi[3] = {-RT_INF, -RT_INF, -RT_INF};
                    float Rlo[3] = {RT_INF, RT_INF, RT_INF}, Rhi[3] = {-RT_INF, -RT_INF, -RT_INF};
                    int lc = 0, rc = 0;
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        const bool valid = j < cnt;
                        const bool inL = valid && ((tb[j] >> (4 * ax)) & 15) <= spl;
                        const bool inR = valid && !inL;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            Llo[k] = fminf(Llo[k], inL ? tlo[j][k] : RT_INF);
                            Lhi[k] = fmaxf(Lhi[k], inL ? thi[j][k] : -RT_INF);
                            Rlo[k] = fminf(Rlo[k], inR ? tlo[j][k] : RT_INF);
                            Rhi[k] = fmaxf(Rhi[k], inR ? thi[j][k] : -RT_INF);
                        }
                        lc += inL ? 1 : 0;
                        rc += inR ? 1 : 0;
                    }
                    if (i < cnt && rc > 0 && depth < BB_MAX_DEPTH) {
                        const float cst = (1.0f + (50.0f * (float)lc) * bb_area6(Llo, Lhi)) + (50.0f * (float)rc) * bb_area6(Rlo, Rhi);
                        if (cst < leaf_cost && (cst < bcost || (cst == bcost && ax == bax && spl < bsp))) {
                            bcost = cst;
                            bax = ax;
                            bsp = spl;
                            bnl = lc;
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                BLlo[k] = Llo[k]; BLhi[k] = Lhi[k]; BRlo[k] = Rlo[k]; BRhi[k] = Rhi[k];
                            }
                        }
                    }
                }
            }
            if (!(bcost < RT_INF)) {
                bs_write_entry(sn, id, nlo, nhi, -1, a, cnt, 0);  // leaf
                return;
            }
            const int c0 = atomicAdd(&l_cnt[0], 2), rec = atomicAdd(&l_cnt[1], 1);
            atomicMax(&l_cnt[2], depth + 1);
            bs_write_entry(sn, id, nlo, nhi, c0, a, cnt, rec);
            const float ca = bs_sel(bax, clo[0], clo[1], clo[2]), cb2 = bs_sel(bax, chi[0], chi[1], chi[2]);
            const float plane = ca + (cb2 - ca) * ((float)(bsp + 1) / (float)BB_BINS);
            int li = 0, ri = 0;  // stable partition of the node's positions
#pragma unroll
            for (int i = 0; i < TN; i++) {
                if (i < cnt) {
                    const bool right = ((tb[i] >> (4 * bax)) & 15) > bsp;
                    const int dest = a + (right ? bnl + ri : li);
                    li += right ? 0 : 1;
                    ri += right ? 1 : 0;
                    l_idx[dest] = (uint16_t)slot[i];
                    l_leaf[dest] = (uint16_t)(c0 + (right ? 1 : 0));
                }
            }
#pragma unroll
            for (int side = 0; side < 2; side++) {
                float lo2[3], hi2[3], cl2[3], ch2[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    lo2[k] = side ? BRlo[k] : BLlo[k];
                    hi2[k] = side ? BRhi[k] : BLhi[k];
                    float ccl = clo[k], cch = chi[k];
                    if (k == bax) {
                        if (side) ccl = plane; else cch = plane;
                    }
                    cl2[k] = fmaxf(ccl, lo2[k]);
                    ch2[k] = fminf(cch, hi2[k]);
                }
                const int a2 = side ? a + bnl : a, cnt2 = side ? cnt - bnl : bnl;
                if (cnt2 >= 2 && depth + 1 < BB_MAX_DEPTH) {
                    if (cnt2 <= 4)
                        bs_queue_put(l_q4[atomicAdd(&l_cnt[3], 1) & (Q4CAP - 1)], cl2, ch2, c0 + side, depth + 1, a2, cnt2);
                    else
                        bs_queue_put(l_q8[atomicAdd(&l_cnt[4], 1) & (Q8CAP - 1)], cl2, ch2, c0 + side, depth + 1, a2, cnt2);
                } else {
                    bs_write_entry(sn, c0 + side, lo2, hi2, -1, a2, cnt2, 0);
                }
            }
        };
        for (;;) {
            bs_sync();
            const int tail4 = __builtin_amdgcn_readfirstlane(l_cnt[3]), tail8 = __builtin_amdgcn_readfirstlane(l_cnt[4]);
            if (TINY > 4 && tail8 > head8)
                batch(std::integral_constant<int, (TINY > 4 ? TINY : 1)>{}, l_q8, Q8CAP - 1, head8, tail8);
            else if (tail4 > head4)
                batch(std::integral_constant<int, 4>{}, l_q4, Q4CAP - 1, head4, tail4);
            else
                break;
        }
        bs_sync();
        m = __builtin_amdgcn_readfirstlane(l_cnt[0]);
        nrec = __builtin_amdgcn_readfirstlane(l_cnt[1]);
        maxd = __builtin_amdgcn_readfirstlane(l_cnt[2]);
    }

    // ---- ids for the new nodes and records, then copy the subtree out --------------------------------------------------
    if (lane == 0) {
        l_alloc[0] = m > 1 ? atomicAdd(&s.counters[0], m - 1) : 0;
        l_alloc[1] = nrec > 0 ? atomicAdd(&s.counters[1], nrec) : 0;
        if (m > 1) atomicMax(&s.counters[3], maxd);
    }
    __threadfence();
    __syncthreads();
    const int base_node = l_alloc[0], base_rec = l_alloc[1];
    for (int j = lane; j < m; j += 64) {
        const float4 e0 = sn[2 * j], e1 = sn[2 * j + 1];
        const int child = (int)rt_f2u(e0.w);
        const uint32_t pk = rt_f2u(e1.w);
        const int g = j == 0 ? root : base_node + j - 1;
        s.nmin[g] = make_float4(e0.x, e0.y, e0.z, 0.0f);
        s.nmax[g] = make_float4(e1.x, e1.y, e1.z, 0.0f);
        s.nstart[g] = start + (int)(pk & 0xfffu);
        s.ncount[g] = (int)((pk >> 12) & 0xfffu);
        s.nchild[g] = child < 0 ? -1 : base_node + child - 1;
        s.nrec[g] = child < 0 ? -1 : base_rec + (int)(pk >> 24);
    }
    for (int p = lane; p < n; p += 64) {
        const int j = (int)l_leaf[p];
        s.order[cur][start + p] = t_id[l_idx[p]];
        s.node_of[cur][start + p] = j == 0 ? root : base_node + j - 1;
    }
}

// ---- refit: same topology and triangle order, boxes recomputed from new vertex positions ----------------------------
__global__ __launch_bounds__(BB_BLOCK) void k_build_parents(BuildState s, int num_nodes) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= num_nodes) return;
    if (i == 0) s.nparent[0] = -1;
    const int c0 = s.nchild[i];
    if (c0 >= 0) {
        s.nparent[c0] = i;
        s.nparent[c0 + 1] = i;
    }
}

// One lane per node; lanes of leaves recompute the leaf's box from its triangles and walk up: the second lane to
// arrive at an inner node (arrival counter) owns both finished children and continues, the first one stops.
//
// Memory ordering.  The two children of a node are in general finished by lanes on different XCDs, whose L2s are not
// coherent with each other.  The textbook form -- release on the first arriver's counter update, acquire on the second's
// -- makes hipcc emit a whole-L2 write-back (buffer_wbl2 sc1) and invalidate (buffer_inv sc1) around every counter
// update: 59 ms for 15 M nodes with fences, 33 ms per refit with an __ATOMIC_ACQ_REL counter (-DRT_REFIT_ACQ_REL) against 2.3 ms.
// What is needed is much less, because only the six floats of one box travel:
//   * boxes are written with agent-scope atomic stores: global_store_dword ... sc1, which writes through the XCD's L2 to
//     the memory-side coherence point instead of staying dirty in it;
//   * s_waitcnt vmcnt(0) then holds the lane until those stores are acknowledged, i.e. visible device-wide; only after
//     that it issues the arrival counter's fetch_add (an agent-scope RMW, performed at the coherence point, so exactly
//     one of the two arrivers reads 1);
//   * the lane that reads 1 knows the sibling's stores were acknowledged before the sibling's fetch_add was issued; it
//     reads the sibling box with agent-scope atomic loads: global_load_dword ... sc1, which miss its own L2 on purpose
//     and fetch from the coherence point.  A GPU lane issues in order and does not speculate past the branch on the
//     counter's value, so these loads cannot start before the fetch_add has returned.
// The compiler is kept from moving the box accesses across the counter update by the asm statement (a "memory" clobber)
// before it and the signal fence after it.  Everything above is gfx942/gfx950 cache behaviour, not the HIP memory model:
// tests/test_gpu_build.py checks after every refit of the 15.8 M-node tree that each inner box is the exact union of
// its children's, so a toolchain that broke the assumption would fail there first.
__device__ __forceinline__ void refit_store(float4* p, float x, float y, float z) {
    float* f = (float*)p;
    __hip_atomic_store(f + 0, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 1, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 2, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void refit_load(const float4* p, float& x, float& y, float& z) {
    const float* f = (const float*)p;
    x = __hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    z = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(BB_BLOCK) void k_build_refit(BuildState s, int cur, int num_nodes, const float* verts,
                                                          const uint32_t* faces) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= num_nodes || s.nchild[i] >= 0) return;
    float lo[3] = {RT_INF, RT_INF, RT_INF}, hi[3] = {-RT_INF, -RT_INF, -RT_INF};
    const int start = s.nstart[i], end = start + s.ncount[i];
    for (int k = start; k < end; k++) {
        const uint32_t t = s.order[cur][k];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t iv = faces[3 * t + c];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float x = verts[3 * iv + d];
                lo[d] = fminf(lo[d], x);
                hi[d] = fmaxf(hi[d], x);
            }
        }
    }
    int node = i;
    for (;;) {
        refit_store(&s.nmin[node], lo[0], lo[1], lo[2]);
        refit_store(&s.nmax[node], hi[0], hi[1], hi[2]);
        const int p = s.nparent[node];
        if (p < 0) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the box has reached the coherence point before the arrival does
#ifdef RT_REFIT_ACQ_REL
        if (__hip_atomic_fetch_add(&s.nflag[p], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
#else
        if (__hip_atomic_fetch_add(&s.nflag[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
#endif
        __atomic_signal_fence(__ATOMIC_SEQ_CST);  // compiler only: the sibling loads stay behind the counter update
        const int sib = s.nchild[p] + (s.nchild[p] == node ? 1 : 0);
        float x, y, z;
        refit_load(&s.nmin[sib], x, y, z);
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        refit_load(&s.nmax[sib], x, y, z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        node = p;
    }
}

// ---- emission ---------------------------------------------------------------------------------------------------
// the 64-byte inner-node records of the render kernels
__global__ __launch_bounds__(BB_BLOCK) void k_build_emit_recs(BuildState s, int num_nodes, float4* recs) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= num_nodes) return;
    const int c0 = s.nchild[i];
    if (c0 < 0) return;
    int ref[2];
    float4 blo[2], bhi[2];
    for (int k = 0; k < 2; k++) {
        int c = c0 + k;
        blo[k] = s.nmin[c];
        bhi[k] = s.nmax[c];
        ref[k] = s.nchild[c] >= 0 ? s.nrec[c] : (s.ncount[c] > 0 ? ~s.nstart[c] : RT_REF_EMPTY);
    }
    float4* q = recs + (size_t)s.nrec[i] * 4;
    q[0] = make_float4(blo[0].x, blo[0].y, blo[0].z, bhi[0].x);
    q[1] = make_float4(bhi[0].y, bhi[0].z, blo[1].x, blo[1].y);
    q[2] = make_float4(blo[1].z, bhi[1].x, bhi[1].y, bhi[1].z);
    q[3] = make_float4(rt_u2f((uint32_t)ref[0]), rt_u2f((uint32_t)ref[1]), 0.0f, 0.0f);
}

// triangle records in leaf order and the shade records (normal as shaders/normals.glsl:13-19 computes it, colour)
__global__ __launch_bounds__(BB_BLOCK) void k_build_emit_tris(BuildState s, int cur, int n, const float* verts,
                                                              const uint32_t* faces, const float4* face_colours, float4* tris,
                                                              float4* shade, uint32_t* edge_max) {
    int k = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (k >= n) return;
    uint32_t t = s.order[cur][k];
    int node = s.node_of[cur][k];
    uint32_t ia = faces[3 * t], ib = faces[3 * t + 1], ic = faces[3 * t + 2];
    const rt_vec3 a = rt_v3(verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]);
    const rt_vec3 b = rt_v3(verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]);
    const rt_vec3 c = rt_v3(verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]);
    bool last = k == s.nstart[node] + s.ncount[node] - 1;
    float4* q = tris + (size_t)k * 4;
    q[0] = make_float4(a.x, a.y, a.z, rt_u2f(last ? 1u : 0u));
    q[1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, rt_u2f((uint32_t)k));
    q[2] = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.0f);
    q[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // edge_max[0] = max |e1| * |e2|, edge_max[1] = max(|e1|, |e2|) over all triangles, as float bits: non-negative floats
    // order like their bit patterns, and a NaN (above every number) marks a scene with a NaN vertex.  The path kernel's
    // cull pad is derived from the two (cull_pad_of in rt_api.hip).
    {
        const float l1 = rt_sqrt(rt_dot(rt_sub(b, a), rt_sub(b, a))), l2 = rt_sqrt(rt_dot(rt_sub(c, a), rt_sub(c, a)));
        uint32_t m1 = rt_f2u(l1 * l2) & 0x7fffffffu, m2 = rt_f2u(l1) & 0x7fffffffu, m3 = rt_f2u(l2) & 0x7fffffffu;
        m2 = m2 > m3 ? m2 : m3;
        const unsigned long long live = __ballot(1);  // the last wave is partly past the end
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o1 = (uint32_t)__shfl_xor((int)m1, off, 64), o2 = (uint32_t)__shfl_xor((int)m2, off, 64);
            if ((live >> ((threadIdx.x & 63) ^ off)) & 1ull) {
                m1 = m1 > o1 ? m1 : o1;
                m2 = m2 > o2 ? m2 : o2;
            }
        }
        // the maxima only grow, so a (possibly stale) plain read that already shows a value as large makes the atomic
        // unnecessary: after the first few waves almost none is issued
        if ((threadIdx.x & 63) == 0) {
            if (m1 > __hip_atomic_load(&edge_max[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&edge_max[0], m1);
            if (m2 > __hip_atomic_load(&edge_max[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&edge_max[1], m2);
        }
    }
    const rt_vec3 nrm = rt_normalize(rt_cross(rt_sub(a, b), rt_sub(a, c)));
    const float4 col = face_colours[t];
    shade[2 * (size_t)k] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
    shade[2 * (size_t)k + 1] = make_float4(col.x, col.y, col.z, 0.0f);
}

// the same tree in the reference's buffer layout (scene-loader.h:7-15), produced on demand for downloads
__global__ __launch_bounds__(BB_BLOCK) void k_build_emit_ref_nodes(BuildState s, int num_nodes, float* ref_nodes, int32_t* links) {
    int i = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= num_nodes) return;
    float4 lo = s.nmin[i], hi = s.nmax[i];
    int c0 = s.nchild[i];
    float* r = ref_nodes + (size_t)12 * i;
    r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = 0.0f;
    r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = 0.0f;
    r[8] = c0 < 0 ? 1.0f : 0.0f;
    r[9] = c0 < 0 ? (float)s.nstart[i] : (float)c0;
    r[10] = c0 < 0 ? (float)(s.nstart[i] + s.ncount[i] - 1) : (float)(c0 + 1);
    r[11] = 0.0f;
    // the same pair as exact integers (floats stop being exact at 2^24)
    links[2 * (size_t)i] = c0 < 0 ? s.nstart[i] : c0;
    links[2 * (size_t)i + 1] = c0 < 0 ? s.nstart[i] + s.ncount[i] - 1 : c0 + 1;
}

__global__ __launch_bounds__(BB_BLOCK) void k_build_emit_ref_tris(BuildState s, int cur, int n, const float* verts,
                                                                  const uint32_t* faces, const float4* face_colours, float4* v0,
                                                                  float4* v1, float4* v2, float4* colours, uint32_t* src) {
    int k = blockIdx.x * BB_BLOCK + threadIdx.x;
    if (k >= n) return;
    uint32_t t = s.order[cur][k];
    uint32_t ia = faces[3 * t], ib = faces[3 * t + 1], ic = faces[3 * t + 2];
    v0[k] = make_float4(verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2], 0.0f);
    v1[k] = make_float4(verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2], 0.0f);
    v2[k] = make_float4(verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2], 0.0f);
    colours[k] = face_colours[t];
    src[k] = t;
}

#endif  // RT_BUILD_H
