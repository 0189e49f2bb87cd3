// Host-side scene ingest: OBJ reader, SAH BVH builder, serialisation into the
// reference's node / vertex buffers, deterministic triangle colours, camera matrix.
//
// The behaviour follows obj-reader.cpp:13-40, bvh.cpp:24-200, scene-loader.cpp:6-30,
// raytrace.cpp:44-52 and main.cpp:41-51 of the reference so that every output byte is the
// same; the structure is new: triangles are kept as flat records, the tree is emitted
// directly in preorder by an explicit job stack (no pointer tree, no recursion), and
// everything crosses a C ABI (include/rt_host.h).
//
// Build with -ffp-contract=off: the SAH costs decide the topology and must be evaluated
// with separate multiplies and adds, as the reference's x86-64 build does.

#include "rt_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <string>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
    float operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }
};

// glm::min / glm::max: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline V3 vmin(V3 a, V3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

// One record per triangle, the four rows of the reference's mat4x3 (bvh.cpp:151-160):
// centroid, min corner, max corner, vertex indices carried as float + 0.2.
struct TriRec {
    V3 centroid, lo, hi, idx;
    int32_t src;  // position in the OBJ face list (extra bookkeeping, moves with the record)
};

const V3 kBigPos = {1e9f, 1e9f, 1e9f};   // MAX_VERTEX, bvh.h:10
const V3 kBigNeg = {-1e9f, -1e9f, -1e9f}; // MIN_VERTEX, bvh.h:9

// Surface area term of the cost model (bvh.cpp:24-31): 6*(dx*dy + dx*dz + dy*dz).
inline float box_area6(V3 lo, V3 hi) {
    float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return 6.0f * ((dx * dy + dx * dz) + dy * dz);
}

// Cost of splitting records [first,last] at `plane` on `axis` (bvh.cpp:33-52).
float split_cost(const std::vector<TriRec>& recs, int first, int last, int axis, float plane) {
    V3 lmin = kBigPos, rmin = kBigPos, lmax = kBigNeg, rmax = kBigNeg;
    float nl = 0.0f, nr = 0.0f;  // counted in float, as the reference does
    for (int j = first; j <= last; ++j) {
        const TriRec& t = recs[j];
        if (t.centroid[axis] < plane) {
            lmin = vmin(lmin, t.lo);
            lmax = vmax(lmax, t.hi);
            nl += 1.0f;
        } else {
            rmin = vmin(rmin, t.lo);
            rmax = vmax(rmax, t.hi);
            nr += 1.0f;
        }
    }
    // an empty side collapses to a zero-size box at +1e9
    lmax = vmax(lmax, lmin);
    rmax = vmax(rmax, rmin);
    return (1.0f + (50.0f * nl) * box_area6(lmin, lmax)) + (50.0f * nr) * box_area6(rmin, rmax);
}

struct Split {
    int axis;   // -1: keep as leaf
    float plane;
    float cost;
};

// Ternary search per axis, best of three against the leaf cost (bvh.cpp:54-80).
Split choose_split(const std::vector<TriRec>& recs, int first, int last, V3 lo, V3 hi) {
    Split best = {-1, 0.0f, (50.0f * (float)(last - first + 1)) * box_area6(lo, hi)};
    for (int axis = 0; axis < 3; ++axis) {
        float a = lo[axis], b = hi[axis];
        for (int it = 0; it < RT_BVH_SPLIT_ITERATIONS; ++it) {
            // The reference's guard is `abs(minVal - maxVal) > 1e-6f` where the unqualified
            // abs binds to ::abs(int) under g++/libstdc++: the float difference is truncated
            // to int first, so the search only runs while the interval is at least 1.0 wide.
            int width = std::abs((int)(a - b));
            if (!((float)width > 1e-6f)) break;
            float third = (b - a) / 3.0f;
            float p1 = a + third, p2 = b - third;
            float c1 = split_cost(recs, first, last, axis, p1);
            float c2 = split_cost(recs, first, last, axis, p2);
            if (c1 <= c2) b = p2;
            if (c2 <= c1) a = p1;
        }
        float plane = (a + b) / 2.0f;
        float cost = split_cost(recs, first, last, axis, plane);
        // min(SplitInfo,SplitInfo) keeps the earlier one only on strictly smaller cost
        if (!(best.cost < cost)) best = {axis, plane, cost};
    }
    return best;
}

struct Job {
    int first, last, depth;
    int parent;  // node whose child slot waits for this subtree's id (-1 for the root)
    int slot;    // 0 = first child, 1 = second child
};

struct NodeOut {
    V3 lo, hi;
    bool leaf;
    int a, b;  // leaf: first/last triangle (inclusive); inner: child ids
};

void build_tree(std::vector<TriRec>& recs, std::vector<NodeOut>& nodes, rt_host_scene* stats) {
    std::vector<Job> todo;
    todo.push_back({0, (int)recs.size() - 1, 0, -1, 0});
    int leaves = 0, min_leaf = 1000000000, max_leaf = 0, max_depth = 0;
    while (!todo.empty()) {
        Job job = todo.back();
        todo.pop_back();
        int id = (int)nodes.size();  // creation order == preorder id (bvh.cpp:118-120)
        if (job.parent >= 0) (job.slot == 0 ? nodes[job.parent].a : nodes[job.parent].b) = id;
        NodeOut node;
        node.lo = kBigPos;
        node.hi = kBigNeg;
        for (int i = job.first; i <= job.last; ++i) {
            node.lo = vmin(node.lo, recs[i].lo);
            node.hi = vmax(node.hi, recs[i].hi);
        }
        Split sp = choose_split(recs, job.first, job.last, node.lo, node.hi);
        node.leaf = (job.depth == RT_MAX_BVH_DEPTH) || sp.axis < 0;
        node.a = node.b = -1;
        if (job.depth > max_depth) max_depth = job.depth;
        if (node.leaf) {
            node.a = job.first;
            node.b = job.last;
            int n = job.last - job.first + 1;
            ++leaves;
            if (n < min_leaf) min_leaf = n;
            if (n > max_leaf) max_leaf = n;
            nodes.push_back(node);
            continue;
        }
        // stable-for-the-left in-place partition (bvh.cpp:108-114)
        int nleft = 0;
        for (int i = job.first; i <= job.last; ++i) {
            if (recs[i].centroid[sp.axis] < sp.plane) {
                std::swap(recs[job.first + nleft], recs[i]);
                ++nleft;
            }
        }
        nodes.push_back(node);
        int mid = job.first + nleft - 1;
        // right subtree is numbered after the whole left subtree: push it first
        todo.push_back({mid + 1, job.last, job.depth + 1, id, 1});
        todo.push_back({job.first, mid, job.depth + 1, id, 0});
    }
    stats->num_leaves = leaves;
    stats->min_leaf_size = min_leaf;
    stats->max_leaf_size = max_leaf;
    stats->max_depth = max_depth;
}

template <class T>
T* alloc_n(size_t n) {
    return (T*)std::malloc(sizeof(T) * (n ? n : 1));
}

// Read one text line of arbitrary length; returns false at end of file.
bool read_line(FILE* f, std::string& line) {
    line.clear();
    int c;
    bool any = false;
    while ((c = std::fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') return true;
        line.push_back((char)c);
    }
    return any;
}

const char* skip_ws(const char* p) {
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f') ++p;
    return p;
}

// istream-style float extraction: skip whitespace, parse the longest valid prefix.
bool take_float(const char*& p, float* out) {
    p = skip_ws(p);
    if (*p == '\0') return false;
    char* end = nullptr;
    float v = std::strtof(p, &end);
    if (end == p) return false;
    // operator>> does not accept inf/nan/hex spellings; reject them the same way
    for (const char* q = p; q < end; ++q) {
        char ch = *q;
        bool ok = (ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E';
        if (!ok) return false;
    }
    p = end;
    *out = v;
    return true;
}

}  // namespace

extern "C" {

void rt_host_free(void* p) { std::free(p); }

int rt_host_read_obj(const char* path, float** vertices, int32_t* num_vertices, uint32_t** faces,
                     int32_t* num_faces) {
    if (!path || !vertices || !num_vertices || !faces || !num_faces) return RT_HOST_EINVAL;
    FILE* f = std::fopen(path, "rb");
    if (!f) return RT_HOST_EIO;
    std::vector<float> vs;
    std::vector<uint32_t> fs;
    std::string line;
    while (read_line(f, line)) {
        if (line.empty()) continue;
        const char* p = skip_ws(line.c_str());
        const char* tok = p;
        while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
        size_t toklen = (size_t)(p - tok);
        if (toklen == 1 && tok[0] == 'v') {
            float x, y, z;
            if (!take_float(p, &x) || !take_float(p, &y) || !take_float(p, &z)) continue;
            vs.push_back(x);
            vs.push_back(y);
            vs.push_back(z);
        } else if (toklen == 1 && tok[0] == 'f') {
            // the reference applies sscanf to the whole line, so the 'f' must be its first byte
            unsigned i1 = 0, i2 = 0, i3 = 0, n1, n2, n3;
            const char* s = line.c_str();
            if (std::sscanf(s, "f %u//%u %u//%u %u//%u", &i1, &n1, &i2, &n2, &i3, &n3) == 6 ||
                std::sscanf(s, "f %u %u %u", &i1, &i2, &i3) == 3) {
                fs.push_back(i1 - 1u);
                fs.push_back(i2 - 1u);
                fs.push_back(i3 - 1u);
            }
        }
        // "#" comments and every other record type are ignored
    }
    std::fclose(f);
    *num_vertices = (int32_t)(vs.size() / 3);
    *num_faces = (int32_t)(fs.size() / 3);
    *vertices = alloc_n<float>(vs.size());
    *faces = alloc_n<uint32_t>(fs.size());
    if (!vs.empty()) std::memcpy(*vertices, vs.data(), vs.size() * sizeof(float));
    if (!fs.empty()) std::memcpy(*faces, fs.data(), fs.size() * sizeof(uint32_t));
    return RT_HOST_OK;
}

int rt_host_build_scene(const float* vertices, int32_t num_vertices, const uint32_t* faces,
                        int32_t num_faces, rt_host_scene* out) {
    if (!vertices || !faces || !out || num_vertices < 0 || num_faces < 0) return RT_HOST_EINVAL;
    std::memset(out, 0, sizeof(*out));
    if (num_faces == 0) return RT_HOST_EEMPTY;
    for (int64_t i = 0; i < (int64_t)num_faces * 3; ++i)
        if (faces[i] >= (uint32_t)num_vertices) return RT_HOST_EINVAL;

    auto vert = [&](uint32_t i) { return V3{vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]}; };
    std::vector<TriRec> recs((size_t)num_faces);
    for (int i = 0; i < num_faces; ++i) {
        uint32_t ia = faces[3 * i], ib = faces[3 * i + 1], ic = faces[3 * i + 2];
        V3 a = vert(ia), b = vert(ib), c = vert(ic);
        TriRec& r = recs[i];
        r.centroid = {((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f};
        r.lo = vmin(a, vmin(b, c));
        r.hi = vmax(a, vmax(b, c));
        r.idx = {(float)ia + 0.2f, (float)ib + 0.2f, (float)ic + 0.2f};
        r.src = i;
    }
    std::vector<NodeOut> nodes;
    nodes.reserve((size_t)num_faces * 2);
    build_tree(recs, nodes, out);

    out->num_tris = num_faces;
    out->num_nodes = (int32_t)nodes.size();
    out->num_vertices = num_vertices;
    out->nodes = alloc_n<float>(nodes.size() * 12);
    out->triv0 = alloc_n<float>((size_t)num_faces * 4);
    out->triv1 = alloc_n<float>((size_t)num_faces * 4);
    out->triv2 = alloc_n<float>((size_t)num_faces * 4);
    out->tri_src = alloc_n<int32_t>((size_t)num_faces);
    out->links = alloc_n<int32_t>(nodes.size() * 2);
    for (size_t i = 0; i < nodes.size(); ++i) {
        const NodeOut& n = nodes[i];
        float* o = out->nodes + 12 * i;
        o[0] = n.lo.x; o[1] = n.lo.y; o[2] = n.lo.z; o[3] = 0.0f;
        o[4] = n.hi.x; o[5] = n.hi.y; o[6] = n.hi.z; o[7] = 0.0f;
        o[8] = n.leaf ? 1.0f : 0.0f;
        o[9] = (float)n.a;
        o[10] = (float)n.b;
        o[11] = 0.0f;
        out->links[2 * i] = (int32_t)n.a;
        out->links[2 * i + 1] = (int32_t)n.b;
    }
    for (int i = 0; i < num_faces; ++i) {
        // indices come back out of the float row by truncation (bvh.cpp:162-164)
        uint32_t ia = (uint32_t)recs[i].idx.x, ib = (uint32_t)recs[i].idx.y, ic = (uint32_t)recs[i].idx.z;
        const uint32_t id[3] = {ia, ib, ic};
        float* dst[3] = {out->triv0 + 4 * i, out->triv1 + 4 * i, out->triv2 + 4 * i};
        for (int k = 0; k < 3; ++k) {
            if (id[k] >= (uint32_t)num_vertices) return RT_HOST_EINVAL;  // > 2^24 vertices: float row overflow
            V3 p = vert(id[k]);
            dst[k][0] = p.x; dst[k][1] = p.y; dst[k][2] = p.z; dst[k][3] = 0.0f;
        }
        out->tri_src[i] = recs[i].src;
    }
    return RT_HOST_OK;
}

int rt_host_load_scene(const char* obj_path, rt_host_scene* out) {
    float* v = nullptr;
    uint32_t* f = nullptr;
    int32_t nv = 0, nf = 0;
    int rc = rt_host_read_obj(obj_path, &v, &nv, &f, &nf);
    if (rc != RT_HOST_OK) return rc;
    rc = rt_host_build_scene(v, nv, f, nf, out);
    std::free(v);
    std::free(f);
    return rc;
}

void rt_host_free_scene(rt_host_scene* s) {
    if (!s) return;
    std::free(s->nodes);
    std::free(s->triv0);
    std::free(s->triv1);
    std::free(s->triv2);
    std::free(s->tri_src);
    std::free(s->links);
    std::memset(s, 0, sizeof(*s));
}

void rt_host_make_colours(uint32_t seed, int32_t num_tris, float* out) {
    // std::mt19937's output sequence is fixed by the C++ standard; the float mapping below is
    // libstdc++'s generate_canonical<float,24>: one 32-bit draw / 2^32, clamped below 1.
    std::mt19937 gen(seed);
    auto draw = [&gen]() {
        float u = (float)(uint32_t)gen() / 4294967296.0f;
        return u >= 1.0f ? 0.99999994f : u;
    };
    for (int32_t i = 0; i < num_tris; ++i) {
        // The reference writes vec4(dist(gen), dist(gen), dist(gen), 1); g++ evaluates the
        // arguments right to left, so the first draw of each triangle lands in blue.
        float first = draw(), second = draw(), third = draw();
        out[4 * i + 0] = third;
        out[4 * i + 1] = second;
        out[4 * i + 2] = first;
        out[4 * i + 3] = 1.0f;
    }
}

void rt_host_camera_rotation(double pitch, double yaw, float out[9]) {
    const float cp = (float)std::cos(pitch), sp = (float)std::sin(pitch);
    const float nsp = (float)(-std::sin(pitch));
    const float cy = (float)std::cos(yaw), sy = (float)std::sin(yaw);
    const float nsy = (float)(-std::sin(yaw));
    // column-major, columns as the GLM constructors of main.cpp:41-50 receive them
    const float X[9] = {1.0f, 0.0f, 0.0f, 0.0f, cp, nsp, 0.0f, sp, cp};
    const float Y[9] = {cy, 0.0f, sy, 0.0f, 1.0f, 0.0f, nsy, 0.0f, cy};
    // R = Y * X ; R[c][r] = (Y[0][r]*X[c][0] + Y[1][r]*X[c][1]) + Y[2][r]*X[c][2]
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            out[3 * c + r] = (Y[0 + r] * X[3 * c + 0] + Y[3 + r] * X[3 * c + 1]) + Y[6 + r] * X[3 * c + 2];
}

uint64_t rt_host_hash64(const void* data, uint64_t num_bytes) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < num_bytes; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

}  // extern "C"
