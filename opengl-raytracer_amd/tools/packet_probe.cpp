// CPU model of a wave-shared ("packet") traversal of the primary rays of 8x8-pixel tiles, next to the
// reference's per-ray ordered traversal (shaders/raytrace.glsl:127-195) of the same rays.
//
// What it answers before any kernel is written:
//   * how many node visits and triangle tests a 64-ray packet needs per tile (the union over its rays), against the
//     per-ray counts;
//   * whether the packet's results (closest hit distance and triangle) equal the per-ray ones on every lane that the
//     packet does not flag as "order-dependent" -- the flags are what a kernel would use to send a ray back to the
//     per-ray traversal.
//
// A ray's closest hit does not depend on the order its nodes are visited in, except where
//   (a) a triangle reports a hit in front of the entry distance of its own leaf box (its EPS-dilated rim sticks out of
//       the box), so that a cull by distance may or may not hide it depending on what was found first, or
//   (b) two triangles report the same distance bit for bit (the first one found wins).
// The packet flags a lane when it sees (a) or (b) on it, or when it culls a node for that lane by a distance that is
// within `margin` of the lane's best hit at that time (a triangle hidden in there could stick out by that much).
//
//   g++ -O2 -std=c++17 -ffp-contract=off -I../../include packet_probe.cpp -L../csrc -lrt_host -Wl,-rpath,'$ORIGIN/../csrc' -o packet_probe
//   ./packet_probe ../../fixtures/dragon80k.obj 1920 1080 0 0 -5.5
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rt_host.h"
#include "rt_math.h"

struct Scene {
    rt_host_scene hs;
    const float* node(int i) const { return hs.nodes + (size_t)i * 12; }
    bool leaf(int i) const { return fabsf(node(i)[8] - 1.0f) <= RT_EPS; }
    int a(int i) const { return hs.links[2 * i]; }
    int b(int i) const { return hs.links[2 * i + 1]; }
};

static float box_dist(rt_vec3 o, rt_vec3 inv, const float* n) {
    float ax = (n[0] - o.x) * inv.x, ay = (n[1] - o.y) * inv.y, az = (n[2] - o.z) * inv.z;
    float bx = (n[4] - o.x) * inv.x, by = (n[5] - o.y) * inv.y, bz = (n[6] - o.z) * inv.z;
    float tNear = rt_max(rt_max(rt_min(ax, bx), rt_min(ay, by)), rt_min(az, bz));
    float tFar = rt_min(rt_min(rt_max(ax, bx), rt_max(ay, by)), rt_max(az, bz));
    bool hit = tFar >= tNear && tFar > RT_EPS;
    return hit ? (tNear > RT_EPS ? tNear : 0.0f) : RT_INF;
}

static float tri_dist(const Scene& s, int k, rt_vec3 o, rt_vec3 d) {
    const float *v0 = s.hs.triv0 + 4 * (size_t)k, *v1 = s.hs.triv1 + 4 * (size_t)k, *v2 = s.hs.triv2 + 4 * (size_t)k;
    rt_vec3 a = rt_v3(v0[0], v0[1], v0[2]);
    rt_vec3 e1 = rt_sub(rt_v3(v1[0], v1[1], v1[2]), a), e2 = rt_sub(rt_v3(v2[0], v2[1], v2[2]), a);
    rt_vec3 p = rt_cross(d, e2);
    float det = rt_dot(e1, p);
    if (rt_abs(det) < RT_EPS) return RT_INF;
    float inv_det = 1.0f / det;
    rt_vec3 sv = rt_sub(o, a);
    float u = inv_det * rt_dot(sv, p);
    if ((u < 0.0f && rt_abs(u) > RT_EPS) || (u > 1.0f && rt_abs(u - 1.0f) > RT_EPS)) return RT_INF;
    rt_vec3 q = rt_cross(sv, e1);
    float v = inv_det * rt_dot(d, q);
    if ((v < 0.0f && rt_abs(v) > RT_EPS) || (u + v > 1.0f && rt_abs((u + v) - 1.0f) > RT_EPS)) return RT_INF;
    float t = inv_det * rt_dot(e2, q);
    return t > RT_EPS ? t : RT_INF;
}

// rt_paths.h: cull_dist -- the distance, or infinity when the ray's TRUE line (1 / true direction in ti) passes the box
// at more than the pad that `slack` stands for; slack < 0: no cull (primary rays)
static float cull_dist(float dist, rt_vec3 o, rt_vec3 ti, float slack, const float* n) {
    if (slack < 0.0f) return dist;
    const float ax = (n[0] - o.x) * ti.x, ay = (n[1] - o.y) * ti.y, az = (n[2] - o.z) * ti.z;
    const float bx = (n[4] - o.x) * ti.x, by = (n[5] - o.y) * ti.y, bz = (n[6] - o.z) * ti.z;
    const float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return hi + slack < lo ? RT_INF : dist;
}

struct Hit {
    float t;
    int id;
};

// the reference's traversal of one ray; counts node visits (inner) and triangle tests
static Hit trace_ref(const Scene& s, rt_vec3 o, rt_vec3 d, rt_vec3 inv, uint64_t& inner, uint64_t& tris, float slack = -1.0f) {
    const rt_vec3 ti = rt_v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    Hit h = {RT_INF, -1};
    if (box_dist(o, inv, s.node(0)) == RT_INF) return h;
    int st[128];
    float sd[128];
    int sp = 0;
    st[sp] = 0;
    sd[sp++] = box_dist(o, inv, s.node(0));
    while (sp > 0) {
        sp--;
        int n = st[sp];
        if (sd[sp] >= h.t) continue;
        if (s.leaf(n)) {
            for (int k = s.a(n); k <= s.b(n); k++) {
                tris++;
                float t = tri_dist(s, k, o, d);
                if (t < h.t) {
                    h.t = t;
                    h.id = k;
                }
            }
        } else {
            inner++;
            int c1 = s.a(n), c2 = s.b(n);
            float d1 = box_dist(o, inv, s.node(c1)), d2 = box_dist(o, inv, s.node(c2));
            d1 = cull_dist(d1, o, ti, slack, s.node(c1));
            d2 = cull_dist(d2, o, ti, slack, s.node(c2));
            if (d1 < d2) {
                if (d2 < h.t) st[sp] = c2, sd[sp++] = d2;
                if (d1 < h.t) st[sp] = c1, sd[sp++] = d1;
            } else {
                if (d1 < h.t) st[sp] = c1, sd[sp++] = d1;
                if (d2 < h.t) st[sp] = c2, sd[sp++] = d2;
            }
        }
    }
    return h;
}

static bool g_bounce_cull = true;
static bool g_ordered = false;  // dissenting lanes revisit the first child after the second (per-lane order = the reference's)

struct PacketStats {
    uint64_t tiles = 0, inner_steps = 0, tri_steps = 0, pops = 0, inner_lanes = 0, tri_lanes = 0, flagged = 0, lanes = 0, max_sp = 0;
    uint64_t empty_tiles = 0, dissent = 0;
};

struct Entry {
    int node;
    uint64_t mask;
    float d[64];
};

static void trace_packet(const Scene& s, const rt_vec3* o, const rt_vec3* d, const rt_vec3* inv, uint64_t valid, const float* margin,
                         Hit* out, bool* flag, PacketStats& ps, const float* slack = nullptr) {
    float best[64], curd[64];
    int id[64];
    for (int l = 0; l < 64; l++) best[l] = RT_INF, id[l] = -1, flag[l] = false;
    uint64_t mask = 0;
    for (int l = 0; l < 64; l++)
        if (valid >> l & 1) {
            curd[l] = box_dist(o[l], inv[l], s.node(0));
            if (curd[l] != RT_INF) mask |= 1ull << l;
        }
    ps.tiles++;
    if (!mask) ps.empty_tiles++;
    std::vector<Entry> st;
    int cur = mask ? 0 : -1;
    // a lane dropped from a node by distance: would a slightly different best have kept it?
    auto near_miss = [&](int l, float dist) {
        if (dist != RT_INF && dist >= best[l] && dist < best[l] + margin[l]) flag[l] = true;
    };
    while (cur >= 0) {
        if (s.leaf(cur)) {
            for (int k = s.a(cur); k <= s.b(cur); k++) {
                ps.tri_steps++;
                for (int l = 0; l < 64; l++)
                    if (mask >> l & 1) {
                        ps.tri_lanes++;
                        float t = tri_dist(s, k, o[l], d[l]);
                        if (t < best[l]) {
                            if (t < curd[l]) flag[l] = true;  // (a) in front of its own leaf box
                            best[l] = t;
                            id[l] = k;
                        } else if (t == best[l] && t != RT_INF) {
                            flag[l] = true;  // (b) exact tie
                        }
                    }
            }
            cur = -1;
        } else {
            ps.inner_steps++;
            ps.inner_lanes += __builtin_popcountll(mask);
            const int c1 = s.a(cur), c2 = s.b(cur);
            float d1[64], d2[64];
            uint64_t w1 = 0, w2 = 0;
            int pref1 = 0, pref2 = 0;
            for (int l = 0; l < 64; l++)
                if (mask >> l & 1) {
                    d1[l] = box_dist(o[l], inv[l], s.node(c1));
                    d2[l] = box_dist(o[l], inv[l], s.node(c2));
                    if (slack) {
                        const rt_vec3 ti = rt_v3(1.0f / d[l].x, 1.0f / d[l].y, 1.0f / d[l].z);
                        d1[l] = cull_dist(d1[l], o[l], ti, slack[l], s.node(c1));
                        d2[l] = cull_dist(d2[l], o[l], ti, slack[l], s.node(c2));
                    }
                    if (d1[l] < best[l]) w1 |= 1ull << l; else near_miss(l, d1[l]);
                    if (d2[l] < best[l]) w2 |= 1ull << l; else near_miss(l, d2[l]);
                    if ((w1 | w2) >> l & 1) (d1[l] < d2[l] ? pref1 : pref2)++;
                }
            if (w1 && w2) {
                const bool first1 = pref1 > pref2;
                uint64_t wf = first1 ? w1 : w2, ws = first1 ? w2 : w1;
                if (g_ordered) {
                    // lanes that want both children but the other one first: they take the second child with everybody
                    // else and come back for the first one afterwards
                    uint64_t dis = 0;
                    for (int l = 0; l < 64; l++)
                        if ((w1 & w2) >> l & 1) {
                            const bool lane_first1 = d1[l] < d2[l];
                            if (lane_first1 != first1) dis |= 1ull << l;
                        }
                    if (dis) {
                        Entry e0;
                        e0.node = first1 ? c1 : c2;
                        e0.mask = dis;
                        memcpy(e0.d, first1 ? d1 : d2, sizeof e0.d);
                        st.push_back(e0);
                        wf &= ~dis;
                        ps.dissent++;
                    }
                }
                Entry e;
                e.node = first1 ? c2 : c1;
                e.mask = ws;
                memcpy(e.d, first1 ? d2 : d1, sizeof e.d);
                st.push_back(e);
                ps.max_sp = std::max<uint64_t>(ps.max_sp, st.size());
                if (wf) {
                    cur = first1 ? c1 : c2;
                    mask = wf;
                    memcpy(curd, first1 ? d1 : d2, sizeof curd);
                } else {
                    cur = -1;
                }
            } else if (w1 | w2) {
                cur = w1 ? c1 : c2;
                mask = w1 | w2;
                memcpy(curd, w1 ? d1 : d2, sizeof curd);
            } else {
                cur = -1;
            }
        }
        while (cur < 0 && !st.empty()) {
            Entry& e = st.back();
            ps.pops++;
            uint64_t m = 0;
            for (int l = 0; l < 64; l++)
                if (e.mask >> l & 1) {
                    if (e.d[l] < best[l]) m |= 1ull << l; else near_miss(l, e.d[l]);
                }
            if (m) {
                cur = e.node;
                mask = m;
                memcpy(curd, e.d, sizeof curd);
            }
            st.pop_back();
        }
    }
    for (int l = 0; l < 64; l++) out[l] = {best[l], id[l]};
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s scene.obj width height camx camy camz [pitch yaw] [margin_scale] [ordered]\n", argv[0]);
        return 2;
    }
    Scene s;
    if (rt_host_load_scene(argv[1], &s.hs) != 0) {
        fprintf(stderr, "cannot load %s\n", argv[1]);
        return 1;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const float cam[3] = {(float)atof(argv[4]), (float)atof(argv[5]), (float)atof(argv[6])};
    float rot[9];
    rt_host_camera_rotation(argc > 8 ? atof(argv[7]) : 0.0, argc > 8 ? atof(argv[8]) : 0.0, rot);
    const double margin_scale = argc > 9 ? atof(argv[9]) : 1.0;
    g_ordered = argc > 10 && atoi(argv[10]) != 0;
    g_bounce_cull = !(argc > 11 && atoi(argv[11]) == 0);
    const float fov = (float)(90.0 * M_PI / 180.0);
    const float pw = (((float)tan((double)(fov / 2.0f)) * 0.1f) * 2.0f) / (float)W;

    // scene measures of the cull pad (rt_api.hip: cull_pad_of) with D stretched to the camera
    double m1 = 0, m2 = 0;
    for (int k = 0; k < s.hs.num_tris; k++) {
        const float *v0 = s.hs.triv0 + 4 * k, *v1 = s.hs.triv1 + 4 * k, *v2 = s.hs.triv2 + 4 * k;
        double e1 = 0, e2 = 0;
        for (int c = 0; c < 3; c++) e1 += (double)(v1[c] - v0[c]) * (v1[c] - v0[c]), e2 += (double)(v2[c] - v0[c]) * (v2[c] - v0[c]);
        e1 = sqrt(e1), e2 = sqrt(e2);
        m1 = std::max(m1, e1 * e2);
        m2 = std::max(m2, std::max(e1, e2));
    }
    double D = 0;
    for (int c = 0; c < 3; c++) {
        double lo = s.node(0)[c], hi = s.node(0)[4 + c];
        double far = std::max(std::max(fabs(cam[c] - lo), fabs(cam[c] - hi)), hi - lo);
        D += far * far;
    }
    D = sqrt(D);
    const double u = 1.0 / 16777216.0, E = 1e-4;
    const double slope = 4.0 * ((u / E) * m1 * 29.0 + 2.0 * u) + 8.0 * u;
    const double near = 4.0 * ((u / E) * m1 * 26.5 * m2 + 2.0 * u * 4.1 * m2) + 4.0 * E * m2 + u * m2;
    const double pad = slope * D + near;
    double Ds = 0;
    for (int c = 0; c < 3; c++) Ds += ((double)s.node(0)[4 + c] - s.node(0)[c]) * ((double)s.node(0)[4 + c] - s.node(0)[c]);
    const double pad_scene = slope * sqrt(Ds) + near;
    printf("scene %s: %d tris %d nodes depth %d; m1 %.4g m2 %.4g D %.4g pad %.4g (margin scale %.3g)\n", argv[1], s.hs.num_tris,
           s.hs.num_nodes, s.hs.max_depth, m1, m2, D, pad, margin_scale);

    PacketStats ps, bps;
    uint64_t b_ref_inner = 0, b_ref_tris = 0, b_rays = 0, b_hits = 0, b_mism = 0;
    uint64_t ref_inner = 0, ref_tris = 0, rays = 0, hits = 0, mism_unflagged = 0, mism_flagged = 0, flagged = 0;
    const int tx = (W + 7) / 8, ty = (H + 7) / 8;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : ref_inner, ref_tris, rays, hits, mism_unflagged, mism_flagged, flagged, b_ref_inner, b_ref_tris, b_rays, b_hits, b_mism)
    for (int tile = 0; tile < tx * ty; tile++) {
        rt_vec3 o[64], d[64], inv[64];
        float margin[64];
        uint64_t valid = 0;
        Hit ref[64], pk[64];
        bool flag[64];
        for (int l = 0; l < 64; l++) {
            const int gx = (tile % tx) * 8 + (l & 7), gy = (tile / tx) * 8 + (l >> 3);
            if (gx >= W || gy >= H) continue;
            valid |= 1ull << l;
            rt_vec3 local = rt_v3(((float)gx - (float)W / 2.0f) * pw, ((float)gy - (float)H / 2.0f) * pw, 0.1f);
            d[l] = rt_normalize(rt_mat3_mul(rot, local));
            o[l] = rt_v3(cam[0], cam[1], cam[2]);
            inv[l] = rt_v3(1.0f / d[l].x, 1.0f / d[l].y, 1.0f / d[l].z);
            const float widest = rt_max(rt_max(rt_abs(inv[l].x), rt_abs(inv[l].y)), rt_abs(inv[l].z));
            margin[l] = (float)(2.0 * pad * 1.001 * margin_scale) * widest;
            ref[l] = trace_ref(s, o[l], d[l], inv[l], ref_inner, ref_tris);
            rays++;
            hits += ref[l].id >= 0;
        }
        PacketStats local_ps;
        trace_packet(s, o, d, inv, valid, margin, pk, flag, local_ps);
#pragma omp critical
        {
            ps.tiles += local_ps.tiles, ps.inner_steps += local_ps.inner_steps, ps.tri_steps += local_ps.tri_steps, ps.pops += local_ps.pops;
            ps.inner_lanes += local_ps.inner_lanes, ps.tri_lanes += local_ps.tri_lanes, ps.empty_tiles += local_ps.empty_tiles;
            ps.max_sp = std::max(ps.max_sp, local_ps.max_sp);
            ps.dissent += local_ps.dissent;
        }
        for (int l = 0; l < 64; l++)
            if (valid >> l & 1) {
                const bool same = rt_f2u(pk[l].t) == rt_f2u(ref[l].t) && pk[l].id == ref[l].id;
                flagged += flag[l];
                if (!same) (flag[l] ? mism_flagged : mism_unflagged)++;
            }
        // first bounce of the lanes that hit: origin = hit point, slab tests with the primary ray's stale 1/dir
        // (raytrace.glsl:247-254), triangle tests and the cull with a uniformly drawn direction on the normal's side
        rt_vec3 bo[64], bd[64];
        float bslack[64], bmargin[64];
        uint64_t bvalid = 0;
        Hit bref[64], bpk[64];
        for (int l = 0; l < 64; l++) {
            bmargin[l] = 0.0f;
            if (!(valid >> l & 1) || ref[l].id < 0) continue;
            bvalid |= 1ull << l;
            bo[l] = rt_add(o[l], rt_scale(d[l], ref[l].t));
            const int k = ref[l].id;
            const float *v0 = s.hs.triv0 + 4 * (size_t)k, *v1 = s.hs.triv1 + 4 * (size_t)k, *v2 = s.hs.triv2 + 4 * (size_t)k;
            const rt_vec3 a = rt_v3(v0[0], v0[1], v0[2]), b = rt_v3(v1[0], v1[1], v1[2]), c = rt_v3(v2[0], v2[1], v2[2]);
            rt_vec3 n = rt_normalize(rt_cross(rt_sub(a, b), rt_sub(a, c)));
            if (rt_dot(n, d[l]) > 0.0f) n = rt_neg(n);
            uint32_t seed = (uint32_t)(tile * 64 + l) * 9781u + 77u;
            rt_vec3 r;
            for (;;) {  // rejection-sampled uniform direction
                r = rt_v3(2.0f * rt_rand(&seed) - 1.0f, 2.0f * rt_rand(&seed) - 1.0f, 2.0f * rt_rand(&seed) - 1.0f);
                const float q = rt_dot(r, r);
                if (q > 1e-4f && q <= 1.0f) break;
            }
            r = rt_normalize(r);
            if (rt_dot(n, r) < 0.0f) r = rt_neg(r);
            bd[l] = r;
            const float widest = rt_max(rt_max(rt_abs(1.0f / r.x), rt_abs(1.0f / r.y)), rt_abs(1.0f / r.z));
            bslack[l] = g_bounce_cull ? (float)(2.0 * pad_scene * 1.001) * widest : -1.0f;
            bref[l] = trace_ref(s, bo[l], bd[l], inv[l], b_ref_inner, b_ref_tris, bslack[l]);
            b_rays++;
            b_hits += bref[l].id >= 0;
        }
        if (bvalid) {
            PacketStats bl;
            bool bflag[64];
            trace_packet(s, bo, bd, inv, bvalid, bmargin, bpk, bflag, bl, bslack);
#pragma omp critical
            {
                bps.tiles += bl.tiles, bps.inner_steps += bl.inner_steps, bps.tri_steps += bl.tri_steps, bps.pops += bl.pops;
                bps.inner_lanes += bl.inner_lanes, bps.tri_lanes += bl.tri_lanes;
                bps.max_sp = std::max(bps.max_sp, bl.max_sp);
                bps.dissent += bl.dissent;
            }
            for (int l = 0; l < 64; l++)
                if (bvalid >> l & 1) b_mism += !(rt_f2u(bpk[l].t) == rt_f2u(bref[l].t) && bpk[l].id == bref[l].id);
        }
    }
    printf("rays %llu, hit fraction %.4f\n", (unsigned long long)rays, (double)hits / rays);
    printf("per ray (reference order): %.2f node visits, %.2f triangle tests\n", (double)ref_inner / rays, (double)ref_tris / rays);
    printf("per tile (packet):         %.2f node visits (%.1f lanes live), %.2f triangle tests (%.1f lanes live), %.2f pops; "
           "%.1f %% of tiles miss the root box; deepest packet stack %llu\n",
           (double)ps.inner_steps / ps.tiles, (double)ps.inner_lanes / std::max<uint64_t>(1, ps.inner_steps), (double)ps.tri_steps / ps.tiles,
           (double)ps.tri_lanes / std::max<uint64_t>(1, ps.tri_steps), (double)ps.pops / ps.tiles, 100.0 * ps.empty_tiles / ps.tiles,
           (unsigned long long)ps.max_sp);
    printf("per-lane equivalent:       %.2f node visits per 64 rays = %.2f x the packet's\n", 64.0 * ref_inner / rays,
           64.0 * ref_inner / rays / ((double)ps.inner_steps / ps.tiles));
    printf("ordered %d: %.3f nodes per tile where lanes disagreed on the order\n", (int)g_ordered, (double)ps.dissent / ps.tiles);
    printf("flagged lanes %llu (%.4f %%), mismatches: %llu on flagged lanes, %llu on UNFLAGGED lanes\n", (unsigned long long)flagged,
           100.0 * flagged / rays, (unsigned long long)mism_flagged, (unsigned long long)mism_unflagged);
    printf("first bounce (cull %d): %llu rays in %llu tiles, hit fraction %.4f; per ray %.2f node visits, %.2f triangle tests\n", (int)g_bounce_cull,
           (unsigned long long)b_rays, (unsigned long long)bps.tiles, (double)b_hits / std::max<uint64_t>(1, b_rays), (double)b_ref_inner / std::max<uint64_t>(1, b_rays),
           (double)b_ref_tris / std::max<uint64_t>(1, b_rays));
    printf("  per tile with hits (packet): %.2f node visits (%.1f lanes live), %.2f triangle tests (%.1f lanes live), %.2f pops, deepest stack %llu, "
           "%.2f disagreements; %.1f bounce rays per such tile; mismatches %llu\n",
           (double)bps.inner_steps / bps.tiles, (double)bps.inner_lanes / std::max<uint64_t>(1, bps.inner_steps), (double)bps.tri_steps / bps.tiles,
           (double)bps.tri_lanes / std::max<uint64_t>(1, bps.tri_steps), (double)bps.pops / bps.tiles, (unsigned long long)bps.max_sp,
           (double)bps.dissent / bps.tiles, (double)b_rays / bps.tiles, (unsigned long long)b_mism);
    rt_host_free_scene(&s.hs);
    return mism_unflagged ? 3 : 0;
}
