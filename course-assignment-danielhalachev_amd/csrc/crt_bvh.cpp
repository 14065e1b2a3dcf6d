// crt_bvh.cpp -- host-side build of the candidate filter (crt_bvh.h): triangle margins, binned-SAH binary build, 4-wide collapse,
// and the two inverse maps the verification step reads (triangle -> reference leaves, mesh -> top-level leaves).
#include "crt_bvh.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr double UNIT_ROUNDOFF = 5.9604644775390625e-8;  // 2^-24

bool is_leaf_link(uint32_t link) { return (link & CRT_LINK_LEAF) != 0 && link != CRT_LINK_END; }

// The nodes of the tree under `root`, ascending (= visit order: both links of a node point forward).  Children of an inner node:
// `link` and -- unless it is the node's own `miss` -- the `miss` of that first child (crt_scene.hip finds the top-level tree the same way).
void collect_tree(const crt_scene_desc *s, uint32_t root, std::vector<bool> &mark, std::vector<uint32_t> &out) {
    std::vector<uint32_t> stack{root};
    while (!stack.empty()) {
        const uint32_t i = stack.back();
        stack.pop_back();
        if (i >= s->n_nodes || mark[i]) continue;
        mark[i] = true;
        out.push_back(i);
        const crt_node &n = s->nodes[i];
        if (is_leaf_link(n.link) || n.link == CRT_LINK_END) continue;
        stack.push_back(n.link);
        const uint32_t c2 = s->nodes[n.link].miss;
        if (c2 != n.miss && c2 != CRT_LINK_END) stack.push_back(c2);
    }
    std::sort(out.begin(), out.end());
}

// How far outside its triangle can the computed hit point of an ACCEPTED hit lie (in the triangle's plane)?
//   Triangle.cpp:37-57 accepts p when, for each edge k, s_k = n . ((v_{k+1} - v_k) x (p - v_k)) is not below -FLT_EPSILON.  In exact
//   arithmetic s_k = |e_k| * (signed in-plane distance of p from the edge's line, positive inside; n is the unit normal of the
//   winding).  The float evaluation errs by at most ~12 u |e_k| |p - v_k| (u = 2^-24: two subtractions, a cross product, a dot
//   product), so an accepted p lies no further than  d_k = (FLT_EPSILON + 64 u |e_k| 2L) / |e_k|  outside edge k (L = longest
//   edge; |p - v_k| <= 2L is checked at the end).  The region {distance to edge k >= -d_k for all k} is a triangle; the margin
//   returned is the largest distance of one of its corners from the corresponding vertex -- a thin corner pushes it far out --,
//   doubled.  false: the bound does not exist or is not small (zero-length edge, needle, a normal that is not this winding's
//   unit normal, non-finite data): such a scene gets no filter.
bool triangle_margin(const crt_triangle &T, double &margin) {
    const double v[3][3] = {{T.v0[0], T.v0[1], T.v0[2]}, {T.v1[0], T.v1[1], T.v1[2]}, {T.v2[0], T.v2[1], T.v2[2]}};
    const double n[3] = {T.nx, T.ny, T.nz};
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(n[k])) return false;
        for (int a = 0; a < 3; a++) if (!std::isfinite(v[k][a])) return false;
    }
    const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nn > 0.999 && nn < 1.001)) return false;
    double e[3][3], len[3], L = 0, lmin = DBL_MAX;
    for (int k = 0; k < 3; k++) {
        for (int a = 0; a < 3; a++) e[k][a] = v[(k + 1) % 3][a] - v[k][a];
        len[k] = std::sqrt(e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2]);
        L = std::max(L, len[k]);
        lmin = std::min(lmin, len[k]);
        if (!(len[k] > 0)) return false;
        if (std::fabs(n[0] * e[k][0] + n[1] * e[k][1] + n[2] * e[k][2]) > 1e-4 * len[k]) return false;  // n is not normal to the plane
    }
    // the winding's normal: e0 x (v2 - v0) must point along n
    const double c[3] = {e[0][1] * -e[2][2] - e[0][2] * -e[2][1], e[0][2] * -e[2][0] - e[0][0] * -e[2][2], e[0][0] * -e[2][1] - e[0][1] * -e[2][0]};
    const double area2 = c[0] * n[0] + c[1] * n[1] + c[2] * n[2];
    if (!(area2 > 1e-12 * L * L)) return false;
    // in-plane frame: u along e0, w = n x u
    const double u[3] = {e[0][0] / len[0], e[0][1] / len[0], e[0][2] / len[0]};
    double w[3] = {n[1] * u[2] - n[2] * u[1], n[2] * u[0] - n[0] * u[2], n[0] * u[1] - n[1] * u[0]};
    const double wl = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int a = 0; a < 3; a++) w[a] /= wl;
    double p[3][2];
    for (int k = 0; k < 3; k++) {
        const double d[3] = {v[k][0] - v[0][0], v[k][1] - v[0][1], v[k][2] - v[0][2]};
        p[k][0] = d[0] * u[0] + d[1] * u[1] + d[2] * u[2];
        p[k][1] = d[0] * w[0] + d[1] * w[1] + d[2] * w[2];
    }
    // offset lines: edge k from p[k] to p[k+1], inward normal m_k (counter-clockwise winding in this frame), pushed out by d_k
    double m[3][2], off[3];
    for (int k = 0; k < 3; k++) {
        const double dx = p[(k + 1) % 3][0] - p[k][0], dy = p[(k + 1) % 3][1] - p[k][1];
        const double l2 = std::sqrt(dx * dx + dy * dy);
        if (!(l2 > 0)) return false;
        m[k][0] = -dy / l2; m[k][1] = dx / l2;
        const double dk = ((double)FLT_EPSILON + 64.0 * UNIT_ROUNDOFF * len[k] * 2.0 * L) / (0.999 * len[k]);
        off[k] = m[k][0] * p[k][0] + m[k][1] * p[k][1] - dk;   // the line m_k . x = off_k
    }
    margin = 0;
    for (int j = 0; j < 3; j++) {  // corner j: lines of the two edges meeting at vertex j, (j + 2) % 3 and j
        const int a = (j + 2) % 3, b = j;
        const double det = m[a][0] * m[b][1] - m[a][1] * m[b][0];
        if (!(std::fabs(det) > 1e-9)) return false;  // a needle: the corner runs away
        const double x = (off[a] * m[b][1] - m[a][1] * off[b]) / det, y = (m[a][0] * off[b] - off[a] * m[b][0]) / det;
        const double dist = std::sqrt((x - p[j][0]) * (x - p[j][0]) + (y - p[j][1]) * (y - p[j][1]));
        if (!std::isfinite(dist)) return false;
        margin = std::max(margin, dist);
    }
    margin *= 2.0;
    return margin <= 0.5 * lmin;  // (also makes |p - v_k| <= 2L hold for every accepted p)
}

// one entry of a verification list: the leaf's box and where the reference collects the hit it stands for
void push_box_entry(std::vector<float> &list, const crt_node &n, uint32_t entry) {
    float eb;
    memcpy(&eb, &entry, 4);
    const float rec[8] = {n.lo[0], n.lo[1], n.lo[2], eb, n.hi[0], n.hi[1], n.hi[2], 0.0f};
    list.insert(list.end(), rec, rec + 8);
}

struct Box { float lo[3], hi[3]; };
void box_empty(Box &b) { for (int a = 0; a < 3; a++) { b.lo[a] = FLT_MAX; b.hi[a] = -FLT_MAX; } }
void box_add(Box &b, const Box &o) { for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); } }
double box_area(const Box &b) {
    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + y * z + z * x);
}

struct BinNode { Box box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; };  // left < 0: a leaf over prims[first, first + count)

struct Builder {
    const std::vector<Box> &pbox;
    std::vector<uint32_t> prims;
    std::vector<BinNode> nodes;
    uint32_t max_depth = 0;
    explicit Builder(const std::vector<Box> &b) : pbox(b) {}

    int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        const int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        max_depth = std::max(max_depth, depth);
        Box bb, cb;
        box_empty(bb); box_empty(cb);
        for (uint32_t i = first; i < first + count; i++) {
            const Box &b = pbox[prims[i]];
            box_add(bb, b);
            for (int a = 0; a < 3; a++) {
                const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
                cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c);
            }
        }
        nodes[id].box = bb;
        if (count <= BVH_LEAF_MAX) { nodes[id].first = first; nodes[id].count = count; return id; }
        // binned SAH over the three axes
        constexpr int BINS = 16;
        double best_cost = DBL_MAX;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Box bins[BINS];
            uint32_t cnt[BINS] = {};
            for (auto &b : bins) box_empty(b);
            const float scale = (float)BINS / ext;
            for (uint32_t i = first; i < first + count; i++) {
                const Box &b = pbox[prims[i]];
                int k = (int)(((0.5f * b.lo[a] + 0.5f * b.hi[a]) - lo) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                cnt[k]++;
                box_add(bins[k], b);
            }
            double right_area[BINS];
            uint32_t right_cnt[BINS];
            Box acc;
            box_empty(acc);
            uint32_t c = 0;
            for (int k = BINS - 1; k > 0; k--) { box_add(acc, bins[k]); c += cnt[k]; right_area[k] = box_area(acc); right_cnt[k] = c; }
            box_empty(acc);
            c = 0;
            for (int k = 1; k < BINS; k++) {
                box_add(acc, bins[k - 1]);
                c += cnt[k - 1];
                if (c == 0 || right_cnt[k] == 0) continue;
                const double cost = box_area(acc) * c + right_area[k] * right_cnt[k];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float lo = cb.lo[a], scale = (float)BINS / (cb.hi[a] - cb.lo[a]);
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](uint32_t p) {
                const Box &b = pbox[p];
                int k = (int)(((0.5f * b.lo[a] + 0.5f * b.hi[a]) - lo) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                return k < best_split;
            });
            mid = (uint32_t)(it - prims.begin());
        } else mid = first + count / 2;          // all centroids coincide: split the list
        if (mid == first || mid == first + count) mid = first + count / 2;
        const int32_t l = build(first, mid - first, depth + 1);
        const int32_t r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
};

}  // namespace

void bvh_build(const crt_scene_desc *s, bool nested_boxes, BvhHost &out) {
    out = BvhHost{};
    auto no = [&](const char *why) { out.ok = false; out.why = why; };
    // ---- the reference's trees: which leaves list a triangle, which top-level leaves list a mesh
    std::vector<bool> mark(s->n_nodes, false);
    std::vector<uint32_t> top_nodes;
    collect_tree(s, s->top_root, mark, top_nodes);
    auto is_range = [](const std::vector<uint32_t> &t) { return t.empty() || t.back() - t.front() + 1 == t.size(); };
    out.trees_are_ranges = is_range(top_nodes);
    {
        std::vector<bool> m2 = mark;
        for (uint32_t m = 0; m < s->n_meshes && out.trees_are_ranges; m++) {
            std::vector<uint32_t> tree;
            if (m2[s->meshes[m].root]) { out.trees_are_ranges = false; break; }
            collect_tree(s, s->meshes[m].root, m2, tree);
            if (!is_range(tree) || tree.front() != s->meshes[m].root) out.trees_are_ranges = false;
        }
    }
    if (!nested_boxes) return no("the reference trees' boxes are not nested");
    if (s->n_leaf_triangles >= (1ull << 31) || s->n_triangles >= (1u << 31)) return no("too many triangles");
    out.tri_mesh.assign(s->n_triangles, NONE);
    std::vector<std::vector<uint32_t>> tri_leaves(s->n_triangles);
    for (uint32_t m = 0; m < s->n_meshes; m++) {
        std::vector<uint32_t> tree;
        if (mark[s->meshes[m].root]) return no("mesh trees share nodes");
        collect_tree(s, s->meshes[m].root, mark, tree);
        for (uint32_t i : tree) {
            const crt_node &n = s->nodes[i];
            if (!is_leaf_link(n.link)) continue;
            uint32_t before = 0;
            for (uint64_t e = n.link & ~CRT_LINK_LEAF; e < s->n_leaf_triangles; e++) {
                const uint32_t t = s->leaf_triangles[e] & ~CRT_ENTRY_LAST;
                // (kernel_bvh.h: bvh_leaf_walk looks a triangle up in a leaf by bisection)
                if (e != (n.link & ~CRT_LINK_LEAF) && t <= before) return no("a leaf's entries do not ascend");
                before = t;
                if (out.tri_mesh[t] != NONE && out.tri_mesh[t] != m) return no("a triangle listed by two meshes");
                out.tri_mesh[t] = m;
                tri_leaves[t].push_back(i);
                tri_leaves[t].push_back((uint32_t)e);
                if (s->leaf_triangles[e] & CRT_ENTRY_LAST) break;
            }
        }
    }
    std::vector<std::vector<uint32_t>> mesh_tops(s->n_meshes);
    for (uint32_t i : top_nodes) {
        const crt_node &n = s->nodes[i];
        if (!is_leaf_link(n.link)) continue;
        for (uint32_t e = n.link & ~CRT_LINK_LEAF; e < s->n_leaf_meshes; e++) {
            const uint32_t m = s->leaf_meshes[e] & ~CRT_ENTRY_LAST;
            mesh_tops[m].push_back(i);
            mesh_tops[m].push_back(e);
            if (s->leaf_meshes[e] & CRT_ENTRY_LAST) break;
        }
    }
    // the scene's largest coordinate, and with it the slack of the overlap predicate (the same float expression on the device)
    for (uint32_t t = 0; t < s->n_triangles; t++)
        for (int a = 0; a < 3; a++)
            out.extent = std::max({out.extent, std::fabs(s->triangles[t].v0[a]), std::fabs(s->triangles[t].v1[a]), std::fabs(s->triangles[t].v2[a])});
    if (!std::isfinite(out.extent)) return no("coordinates out of range");
    out.overlap_eps = out.extent * 0x1p-20f;
    out.tri_leaf_first.assign((size_t)s->n_triangles + 1, 0);
    for (uint32_t t = 0; t < s->n_triangles; t++) {
        const size_t n = tri_leaves[t].size() / 2;
        if (n > BVH_LIST_MAX) {
            // verified by the pruned tree walk: every leaf that lists it must overlap its box, or the walk could miss that leaf
            const crt_triangle &T = s->triangles[t];
            for (size_t j = 0; j < n; j++) {
                const crt_node &nd = s->nodes[tri_leaves[t][2 * j]];
                for (int a = 0; a < 3; a++) {
                    const float lo = std::min({T.v0[a], T.v1[a], T.v2[a]}) - out.overlap_eps, hi = std::max({T.v0[a], T.v1[a], T.v2[a]}) + out.overlap_eps;
                    if (!(lo <= nd.hi[a] && hi >= nd.lo[a])) return no("a leaf lists a triangle that lies outside its box");
                }
            }
            out.tri_mesh[t] |= BVH_TRI_WALK;
            out.walk_triangles++;
            out.tri_leaf_first[t + 1] = out.tri_leaf_first[t];
            continue;
        }
        out.tri_leaf_first[t + 1] = out.tri_leaf_first[t] + (uint32_t)n;
        for (size_t j = 0; j < n; j++) push_box_entry(out.tri_leaf_list, s->nodes[tri_leaves[t][2 * j]], tri_leaves[t][2 * j + 1]);
    }
    out.mesh_top_first.assign((size_t)s->n_meshes + 1, 0);
    for (uint32_t m = 0; m < s->n_meshes; m++) {
        out.mesh_top_first[m + 1] = out.mesh_top_first[m] + (uint32_t)(mesh_tops[m].size() / 2);
        for (size_t j = 0; j < mesh_tops[m].size() / 2; j++) push_box_entry(out.mesh_top_list, s->nodes[mesh_tops[m][2 * j]], mesh_tops[m][2 * j + 1]);
    }
    // ---- the triangles a ray can be tested against at all, each in a box grown by its margin
    std::vector<Box> pbox(s->n_triangles), vbox(s->n_triangles);
    Builder B(pbox);
    for (uint32_t t = 0; t < s->n_triangles; t++) {
        if (out.tri_mesh[t] == NONE) continue;             // no leaf lists it: the reference never tests it
        const uint32_t m = out.tri_mesh[t] & ~BVH_TRI_WALK;
        if (mesh_tops[m].empty()) continue;
        const crt_triangle &T = s->triangles[t];
        double margin;
        if (!triangle_margin(T, margin)) return no("a triangle without a usable margin (degenerate, or its normal is not its winding's)");
        out.max_margin = std::max(out.max_margin, margin);
        // the kernels recompute the plane offset from the record (Ray.cpp:17): it must be the stored one, bit for bit
        const float plane = -(T.v0[0] * T.nx + T.v0[1] * T.ny + T.v0[2] * T.nz);
        if (memcmp(&plane, &T.plane, 4) != 0) return no("a stored plane offset is not -(v0 . n)");
        Box &b = pbox[t];
        for (int a = 0; a < 3; a++) {
            const double lo = std::min({(double)T.v0[a], (double)T.v1[a], (double)T.v2[a]}) - margin;
            const double hi = std::max({(double)T.v0[a], (double)T.v1[a], (double)T.v2[a]}) + margin;
            b.lo[a] = std::nextafter((float)lo, -FLT_MAX);   // (float)x rounds to nearest: one more step outwards
            b.hi[a] = std::nextafter((float)hi, FLT_MAX);
            if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) return no("coordinates out of range");
        }
        // ... and, for the miss check, the box of everything the reference can reach this triangle through: the leaves listing it
        Box &v = vbox[t];
        v = b;
        for (size_t j = 0; j < tri_leaves[t].size(); j += 2) {
            const crt_node &nd = s->nodes[tri_leaves[t][j]];
            for (int a = 0; a < 3; a++) { v.lo[a] = std::min(v.lo[a], nd.lo[a]); v.hi[a] = std::max(v.hi[a], nd.hi[a]); }
        }
        for (int a = 0; a < 3; a++) if (!std::isfinite(v.lo[a]) || !std::isfinite(v.hi[a])) return no("coordinates out of range");
        B.prims.push_back(t);
    }
    if (B.prims.size() >= (1u << 24)) return no("too many triangles for the leaf links");   // (also keeps 48-byte entry offsets in 32 bits: kernel_bvh.h, bvh_entry)
    auto ids_of = [&](uint32_t t) { return t | ((s->meshes[out.tri_mesh[t] & ~BVH_TRI_WALK].flags & 1u) ? BVH_ID_REFRACTIVE : 0u); };
    auto empty_node = [] {
        BvhNode N;
        for (int c = 0; c < 4; c++) {
            N.lox[c] = N.loy[c] = N.loz[c] = FLT_MAX;
            N.hix[c] = N.hiy[c] = N.hiz[c] = -FLT_MAX;
            N.child[c] = BVH_EMPTY; N.pad[c] = 0;
        }
        return N;
    };
    if (B.prims.empty()) {
        out.nodes.push_back(empty_node()); out.wide_depth = 1;
        out.vnodes.assign(24, 0.0f); out.cones.assign(16, 0.0f);
        out.ok = true;
        return;
    }
    const int32_t root = B.build(0, (uint32_t)B.prims.size(), 0);
    out.max_depth = B.max_depth;
    // ---- what the miss check reads: per node of the binary build, the box of its triangles' leaf unions and a cone around their normals
    struct Agg { Box v; double ax[3]; double theta; };   // theta: half-angle (pi: no bound)
    std::vector<Agg> agg(B.nodes.size());
    {
        constexpr double PI = 3.14159265358979323846;
        auto angle = [](const double a[3], const double b[3]) {
            const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            return std::acos(std::max(-1.0, std::min(1.0, d)));
        };
        // post-order without recursion: children have larger indices than their parent (Builder::build), so a reverse sweep does it
        for (size_t i = B.nodes.size(); i-- > 0;) {
            const BinNode &n = B.nodes[i];
            Agg &g = agg[i];
            box_empty(g.v);
            double sum[3] = {0, 0, 0};
            if (n.left < 0) {
                for (uint32_t k = n.first; k < n.first + n.count; k++) {
                    const crt_triangle &T = s->triangles[B.prims[k]];
                    box_add(g.v, vbox[B.prims[k]]);
                    sum[0] += T.nx; sum[1] += T.ny; sum[2] += T.nz;
                }
            } else {
                box_add(g.v, agg[n.left].v); box_add(g.v, agg[n.right].v);
                for (int a = 0; a < 3; a++) sum[a] = agg[n.left].ax[a] + agg[n.right].ax[a];
            }
            const double len = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
            if (!(len > 1e-9)) { g.ax[0] = 1; g.ax[1] = 0; g.ax[2] = 0; g.theta = PI; continue; }
            for (int a = 0; a < 3; a++) g.ax[a] = sum[a] / len;
            g.theta = 0;
            if (n.left < 0) {
                for (uint32_t k = n.first; k < n.first + n.count; k++) {
                    const crt_triangle &T = s->triangles[B.prims[k]];
                    const double nn[3] = {T.nx, T.ny, T.nz};   // (a unit vector to 1e-3: triangle_margin; the angle does not care)
                    const double l = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                    const double u[3] = {nn[0] / l, nn[1] / l, nn[2] / l};
                    g.theta = std::max(g.theta, angle(g.ax, u));
                }
            } else {
                for (int32_t c : {n.left, n.right}) g.theta = std::max(g.theta, std::min(PI, angle(g.ax, agg[c].ax) + agg[c].theta));
            }
            g.theta = std::min(PI, g.theta + 1e-6);
        }
    }
    // ---- collapse to four children per node; leaves' triangles in leaf order
    auto emit_leaf = [&](const BinNode &n) -> uint32_t {
        const uint32_t first = (uint32_t)out.ids.size();
        for (uint32_t i = n.first; i < n.first + n.count; i++) {
            const uint32_t t = B.prims[i];
            const crt_triangle &T = s->triangles[t];
            const float rec[12] = {T.v0[0], T.v0[1], T.v0[2], T.nx, T.v1[0], T.v1[1], T.v1[2], T.ny, T.v2[0], T.v2[1], T.v2[2], T.nz};
            out.tris.insert(out.tris.end(), rec, rec + 12);
            out.ids.push_back(ids_of(t));
        }
        return BVH_LEAF | ((n.count - 1u) << 24) | first;
    };
    struct Work { int32_t bin; uint32_t node; uint32_t depth; };
    std::vector<Work> todo;
    out.nodes.push_back(empty_node());
    todo.push_back({root, 0, 1});
    for (size_t w = 0; w < todo.size(); w++) {   // breadth-first: the top of the tree in consecutive lines
        const int32_t bin = todo[w].bin;
        const uint32_t at = todo[w].node, depth = todo[w].depth;
        out.wide_depth = std::max(out.wide_depth, depth);
        int32_t kids[4];
        int nk = 0;
        if (B.nodes[bin].left < 0) kids[nk++] = bin;   // (only the root can be a leaf here)
        else { kids[nk++] = B.nodes[bin].left; kids[nk++] = B.nodes[bin].right; }
        while (nk < 4) {
            int pick = -1;
            double pa = -1;
            for (int k = 0; k < nk; k++)
                if (B.nodes[kids[k]].left >= 0 && box_area(B.nodes[kids[k]].box) > pa) { pa = box_area(B.nodes[kids[k]].box); pick = k; }
            if (pick < 0) break;
            const int32_t k0 = kids[pick];
            kids[pick] = B.nodes[k0].left;
            kids[nk++] = B.nodes[k0].right;
        }
        BvhNode N = empty_node();
        float V[24], C[16];
        for (int k = 0; k < 4; k++) {
            V[k] = V[4 + k] = V[8 + k] = FLT_MAX; V[12 + k] = V[16 + k] = V[20 + k] = -FLT_MAX;
            C[k] = 1.0f; C[4 + k] = C[8 + k] = 0.0f; C[12 + k] = 2.0f;
        }
        for (int k = 0; k < nk; k++) {
            const BinNode &c = B.nodes[kids[k]];
            const Agg &g = agg[kids[k]];
            for (int a = 0; a < 3; a++) {
                V[4 * a + k] = g.v.lo[a]; V[12 + 4 * a + k] = g.v.hi[a];
                C[4 * a + k] = (float)g.ax[a];
                out.extent = std::max({out.extent, std::fabs(g.v.lo[a]), std::fabs(g.v.hi[a])});
            }
            // skip the child when |d . axis| > sin(theta + 1e-4): then no normal of the cone is within 1e-4 rad of perpendicular to d
            const double bound = g.theta + 1e-4;
            C[12 + k] = bound >= 1.5707 ? 2.0f : std::nextafter((float)std::sin(bound), 4.0f);
            N.lox[k] = c.box.lo[0]; N.loy[k] = c.box.lo[1]; N.loz[k] = c.box.lo[2];
            N.hix[k] = c.box.hi[0]; N.hiy[k] = c.box.hi[1]; N.hiz[k] = c.box.hi[2];
            for (int a = 0; a < 3; a++) out.extent = std::max({out.extent, std::fabs(c.box.lo[a]), std::fabs(c.box.hi[a])});  // (the margins on top)
            if (c.left < 0) N.child[k] = emit_leaf(c);
            else {
                N.child[k] = (uint32_t)out.nodes.size();
                out.nodes.push_back(empty_node());
                todo.push_back({kids[k], N.child[k], depth + 1});
            }
        }
        out.nodes[at] = N;
        if (out.vnodes.size() < out.nodes.size() * 24) { out.vnodes.resize(out.nodes.size() * 24, 0.0f); out.cones.resize(out.nodes.size() * 16, 0.0f); }
        std::copy(V, V + 24, out.vnodes.begin() + (size_t)at * 24);
        std::copy(C, C + 16, out.cones.begin() + (size_t)at * 16);
    }
    out.vnodes.resize(out.nodes.size() * 24, 0.0f);
    out.cones.resize(out.nodes.size() * 16, 0.0f);
    if (out.nodes.size() >= (1u << 31)) return no("too many nodes");
    out.ok = true;
}

#ifdef CRT_TEST_HOOKS   // (libcrt_hip_test.so only)
// =================================================================================================================================
// crt_bvh_selftest: the filter's two promises, checked on the HOST against brute force (tests/test_bvh_filter.py; no GPU involved).
// The tests below are the host's copies of kernel_bvh.h's (bvh_ray_setup / bvh_child_test, bvh_line_setup / bvh_line_test, the cone
// test, bvh_triangle): the same expressions, compiled without contraction like everything else here.
//   (1) every triangle the reference's test ACCEPTS with a finite distance for a ray is reached by the walk of the hierarchy with
//       the conservative box test, whatever distance bound the walk has learnt so far (the bound used: that very distance);
//   (2) every triangle it accepts with ANY distance (an infinite or NaN one included) and whose reference leaves the ray's line can
//       pass is reached by the miss check's walk (reach boxes, normal cones).
// out: {rays, accepted finite hits, of which not reached (1), accepted hits of any distance, of which not reached (2), nodes visited
//       by (1), nodes visited by (2), structural errors}
namespace {

struct HRay { float ox, oy, oz, dx, dy, dz; };
struct HBox { float ix, iy, iz, cpx, cpy, cpz, cmx, cmy, cmz; };
inline float hdot3(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }

HBox h_ray_setup(const HRay &R, float extent, float tiny, float sub) {
    const float rho = (extent + std::fmax(std::fmax(std::fabs(R.ox), std::fabs(R.oy)), std::fabs(R.oz))) * 0x1p-16f;
    const float dx = std::fabs(R.dx) < tiny ? std::copysign(sub, R.dx) : R.dx, dy = std::fabs(R.dy) < tiny ? std::copysign(sub, R.dy) : R.dy,
                dz = std::fabs(R.dz) < tiny ? std::copysign(sub, R.dz) : R.dz;
    HBox B;
    B.ix = 1.0f / dx; B.iy = 1.0f / dy; B.iz = 1.0f / dz;
    B.cpx = -((R.ox + rho) * B.ix); B.cmx = -((R.ox - rho) * B.ix);
    B.cpy = -((R.oy + rho) * B.iy); B.cmy = -((R.oy - rho) * B.iy);
    B.cpz = -((R.oz + rho) * B.iz); B.cmz = -((R.oz - rho) * B.iz);
    return B;
}
bool h_box_test(const HBox &B, const float *lo, const float *hi, bool segment, float tmax) {
    const float ax = __builtin_fmaf(lo[0], B.ix, B.cpx), bx = __builtin_fmaf(hi[0], B.ix, B.cmx);
    const float ay = __builtin_fmaf(lo[1], B.iy, B.cpy), by = __builtin_fmaf(hi[1], B.iy, B.cmy);
    const float az = __builtin_fmaf(lo[2], B.iz, B.cpz), bz = __builtin_fmaf(hi[2], B.iz, B.cmz);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (segment) { tn = fmaxf(tn, 0.0f); tf = fminf(tf, tmax); }
    return !(__builtin_fmaf(-std::fabs(tn), 0x1p-20f, tn) > __builtin_fmaf(std::fabs(tf), 0x1p-20f, tf));
}
bool h_triangle(const HRay &R, bool primary, const float *T, float &t) {   // T: {v0, nx} {v1, ny} {v2, nz}
    const float nx = T[3], ny = T[7], nz = T[11];
    const float nd = hdot3(R.dx, R.dy, R.dz, nx, ny, nz);
    const float plane = -hdot3(T[0], T[1], T[2], nx, ny, nz);
    t = -(hdot3(nx, ny, nz, R.ox, R.oy, R.oz) + plane) / nd;
    const float px = R.ox + R.dx * t, py = R.oy + R.dy * t, pz = R.oz + R.dz * t;
    float s[3];
    for (int k = 0; k < 3; k++) {
        const float *a = T + 4 * k, *b = T + 4 * ((k + 1) % 3);
        const float ex = b[0] - a[0], ey = b[1] - a[1], ez = b[2] - a[2], cx = px - a[0], cy = py - a[1], cz = pz - a[2];
        s[k] = hdot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx);
    }
    return !(primary && nd >= 0) && !(t < 0) && !(s[0] < -FLT_EPSILON) && !(s[1] < -FLT_EPSILON) && !(s[2] < -FLT_EPSILON);
}

}  // namespace

extern "C" int crt_bvh_selftest(const crt_scene_desc *s, const float *rays, uint32_t n_rays, int primary, uint64_t out[8]) {
    if (!s || !rays || !out) return CRT_ERR_INVALID;
    for (int k = 0; k < 8; k++) out[k] = 0;
    BvhHost H;
    bvh_build(s, true, H);
    if (!H.ok) return CRT_ERR_INVALID;
    const size_t n_entries = H.ids.size(), n_nodes = H.nodes.size();
    // structure: every child reference inside its array, every entry in exactly one leaf
    std::vector<uint32_t> seen(n_entries, 0);
    for (size_t i = 0; i < n_nodes; i++)
        for (int c = 0; c < 4; c++) {
            const uint32_t ch = H.nodes[i].child[c];
            if (ch == BVH_EMPTY) continue;
            if (ch & BVH_LEAF) {
                const uint32_t first = ch & 0x00FFFFFFu, n = ((ch >> 24) & 0x7Fu) + 1u;
                if (first + n > n_entries) { out[7]++; continue; }
                for (uint32_t k = 0; k < n; k++) seen[first + k]++;
            } else if (ch >= n_nodes || ch <= i) out[7]++;
        }
    for (size_t e = 0; e < n_entries; e++) if (seen[e] != 1) out[7]++;
    if (H.vnodes.size() != n_nodes * 24 || H.cones.size() != n_nodes * 16 || H.tris.size() != n_entries * 12) out[7]++;
    if (out[7]) return CRT_OK;
    // the reach of entry e: can the ray's line pass one of the reference leaves listing its triangle?  (the reference's own slab test,
    // BoundingBox.h:85-108, on those boxes -- what kernel_bvh.h: bvh_verify decides)
    auto ref_slab = [](const HRay &R, const float *lo, const float *hi) {
        float t0 = -FLT_MAX, t1 = FLT_MAX;
        const float o[3] = {R.ox, R.oy, R.oz}, d[3] = {R.dx, R.dy, R.dz};
        for (int a = 0; a < 3; a++) {
            if (std::fabs(d[a]) < FLT_EPSILON) { if (o[a] < lo[a] || o[a] > hi[a]) return false; continue; }
            const float inv = 1.0f / d[a];
            float tn = (lo[a] - o[a]) * inv, tf = (hi[a] - o[a]) * inv;
            if (tn > tf) std::swap(tn, tf);
            t0 = (t0 < tn) ? tn : t0; t1 = (tf < t1) ? tf : t1;
            if (t0 > t1) return false;
        }
        return true;
    };
    std::vector<uint32_t> stack;
    std::vector<char> reached(n_entries);
    for (uint32_t r = 0; r < n_rays; r++) {
        HRay R{rays[6 * r], rays[6 * r + 1], rays[6 * r + 2], rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]};
        out[0]++;
        // ---- (1) finite accepted hits against the segment walk, each with its own distance as the bound
        const HBox B = h_ray_setup(R, H.extent, 1e-12f, 1e-12f);
        for (size_t e = 0; e < n_entries; e++) {
            float t;
            if (!h_triangle(R, primary != 0, &H.tris[12 * e], t) || !(t < INFINITY)) continue;
            out[1]++;
            bool found = false;
            stack.assign(1, 0u);
            while (!stack.empty() && !found) {
                const uint32_t cur = stack.back();
                stack.pop_back();
                if (cur & BVH_LEAF) {
                    const uint32_t first = cur & 0x00FFFFFFu, n = ((cur >> 24) & 0x7Fu) + 1u;
                    if (e >= first && e < first + n) found = true;
                    continue;
                }
                out[5]++;
                const BvhNode &N = H.nodes[cur];
                for (int c = 0; c < 4; c++) {
                    if (N.child[c] == BVH_EMPTY) continue;
                    const float lo[3] = {N.lox[c], N.loy[c], N.loz[c]}, hi[3] = {N.hix[c], N.hiy[c], N.hiz[c]};
                    if (h_box_test(B, lo, hi, true, t)) stack.push_back(N.child[c]);
                }
            }
            if (!found) out[2]++;
        }
        // ---- (2) accepted hits of any distance against the miss check's walk
        const HBox Lb = h_ray_setup(R, H.extent, 2.0f * FLT_EPSILON, 1e-30f);
        std::fill(reached.begin(), reached.end(), 0);
        stack.assign(1, 0u);
        while (!stack.empty()) {
            const uint32_t cur = stack.back();
            stack.pop_back();
            if (cur & BVH_LEAF) {
                const uint32_t first = cur & 0x00FFFFFFu, n = ((cur >> 24) & 0x7Fu) + 1u;
                for (uint32_t k = 0; k < n; k++) reached[first + k] = 1;
                continue;
            }
            out[6]++;
            const BvhNode &N = H.nodes[cur];
            const float *V = &H.vnodes[(size_t)cur * 24], *C = &H.cones[(size_t)cur * 16];
            for (int c = 0; c < 4; c++) {
                if (N.child[c] == BVH_EMPTY) continue;
                const float lo[3] = {V[c], V[4 + c], V[8 + c]}, hi[3] = {V[12 + c], V[16 + c], V[20 + c]};
                const float sdot = __builtin_fmaf(C[c], R.dx, __builtin_fmaf(C[4 + c], R.dy, C[8 + c] * R.dz));
                if (!(std::fabs(sdot) > C[12 + c]) && h_box_test(Lb, lo, hi, false, 0.0f)) stack.push_back(N.child[c]);
            }
        }
        for (size_t e = 0; e < n_entries; e++) {
            float t;
            if (!h_triangle(R, primary != 0, &H.tris[12 * e], t)) continue;
            if (t < INFINITY) continue;   // (finite ones are (1)'s)
            // would the reference test it?  one of its leaves' boxes must pass the reference's slab test
            const uint32_t tri = H.ids[e] & ~BVH_ID_REFRACTIVE;
            bool visited = false;
            if (H.tri_mesh[tri] & BVH_TRI_WALK) visited = true;   // (no list: count it as visited -- the stricter demand)
            for (uint32_t j = H.tri_leaf_first[tri]; j < H.tri_leaf_first[tri + 1] && !visited; j++)
                visited = ref_slab(R, &H.tri_leaf_list[8 * (size_t)j], &H.tri_leaf_list[8 * (size_t)j + 4]);
            if (!visited) continue;
            out[3]++;
            if (!reached[e]) out[4]++;
        }
    }
    return CRT_OK;
}
#endif  // CRT_TEST_HOOKS
