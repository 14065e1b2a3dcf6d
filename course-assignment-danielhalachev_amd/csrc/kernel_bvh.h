// kernel_bvh.h -- the production walk kernels: candidates from the filter hierarchy (crt_bvh.h), the reference's exact triangle
// test on them, and the verification of every accepted candidate against the reference's own trees.
//
// What has to hold for the frame to be the reference's, float for float (crt_bvh.h has the argument for the candidates):
//   * bvh_child_test can only err towards "pass": the boxes hold every triangle grown by its acceptance margin; a ray is given a
//     slack rho = 2^-16 (largest box coordinate + largest origin coordinate) -- 256 unit roundoffs of the magnitudes involved, where
//     the analysis needs about 20 (hit point o + d t rounded per component; the plane residual of a computed t, 6 u t + 4 sqrt(3) u
//     (2 |o| + |v|)) -- the interval ends are widened by 2^-20 of themselves (their own rounding: ~4 u), and a NaN anywhere passes;
//   * the triangle test is kernel_plan.h's, expression for expression (Ray.cpp:9-31, Triangle.cpp:37-57);
//   * bvh_verify: the reference tests triangle T of mesh M for this ray exactly when it reaches a top-level leaf listing M and a leaf
//     of M's tree listing T, and it reaches a leaf exactly when the leaf's own box passes the exact slab test (nested boxes, monotone
//     test: kernel_heavy.h).  Both lists are walked in visit order up to the first box that passes, whose entry position is the place
//     where the reference first COLLECTS the hit: ties in distance go to the smallest (mesh entry, triangle entry) -- KDTree.cpp:75-86,
//     156-167 take the first collected hit and replace it only by a strictly smaller distance;
//   * shadow rays: occluded == some accepted hit with a finite distance lies within the light's distance (kernel_walk.h), whatever
//     the order; hits beyond the light cannot count, so the walk ends at distance * (1 + 2^-16) (|d| = 1 up to 2 u);
//   * a closest-hit ray WITHOUT a verified finite hit: the reference still gives it a hit if some triangle it tests is accepted with
//     an infinite or NaN distance (KDTree.cpp:75-86: only a finite distance replaces the first collected hit).  Such a distance needs
//     a computed d . n of zero (or so small that the quotient overflows): |d . n| < 4 u.  THE MISS CHECK (bvh_miss_step) looks for
//     exactly those triangles: the same hierarchy, read through the boxes of everything the reference can reach a child's triangles
//     through (the LINE of the ray against them: the reference's slab test knows no t >= 0) and through cones around the children's
//     normals (a child none of whose normals is within 1e-4 rad of perpendicular to d is skipped); what is left is tested with
//     the reference's triangle test and verified like any candidate.  Nothing found -- the all but certain outcome --: the ray is a
//     miss.  Something found: the frame's overflow word is raised (with the guard word: larger queues would not help) and the
//     queue-less reference-order kernel redoes the frame (crt_stats::fallback_frames), the same last resort as a queue overflow;
//   * a ray with non-finite coordinates (the filter would pass everything) condemns the frame the same way.
#pragma once

#include "crt_bvh.h"
#include "kernel_common.h"
#include "kernel_stream.h"
#include "kernel_walk.h"

// The walk's stack: the first BVH_LDS_STACK entries of a lane in LDS (one column per thread: 16 KB per workgroup, so that the level
// kernels and the bulk shadow pass fit on a CU together), the rest -- rarely reached -- in a device buffer (FrameArgs::bvh_spill, one
// column per thread of the largest grid).  SceneArgs::bvh_stack entries in all, sized by crt_create from the hierarchy's depth -- three
// entries per inner node on a path is all a walk can push -- so that no walk outgrows it.
struct BvhStack { uint32_t *lds; uint32_t *spill; uint32_t stride; };   // lds + threadIdx.x; spill + global thread; threads of the grid
__device__ __forceinline__ void bvh_push(const BvhStack &S, uint32_t &sp, const uint32_t v) {
    if (sp < BVH_LDS_STACK) S.lds[sp * BLOCK] = v;
    else S.spill[(size_t)(sp - BVH_LDS_STACK) * S.stride] = v;
    sp++;
}
__device__ __forceinline__ uint32_t bvh_pop(const BvhStack &S, uint32_t &sp) {
    sp--;
    return sp < BVH_LDS_STACK ? S.lds[sp * BLOCK] : S.spill[(size_t)(sp - BVH_LDS_STACK) * S.stride];
}
__device__ __forceinline__ BvhStack bvh_stack_of(const KernelArgs &A, uint32_t *stack_lds, const bool side) {
    BvhStack S;
    S.lds = stack_lds + threadIdx.x;
    S.stride = gridDim.x * BLOCK;
    S.spill = (side ? A.f->bvh_spill_side : A.f->bvh_spill) + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    return S;
}
constexpr float BVH_WIDEN = 0x1p-20f;
// A wave looks after its free lanes -- output of the finished walks, new rays -- when BVH_BATCH of them have gathered (or none walks
// any more): every such round costs the lanes that still walk a shading's worth of time, so the rounds must be few.
constexpr uint32_t BVH_BATCH = 24;
// (level 0 and the bulk shadow pass claim their work indices in chunks: KernelArgs::chunk, crt_tuning::fetch_chunk)
constexpr int BVH_STEPS = 16;        // walk steps between two looks at the lanes' states (measured 4 / 8 / 16 / 32: HW14 2.64 / 2.57 / 2.49 / 2.50 ms, HW12 9.16 / 8.93 / 8.78 / 8.84)
constexpr int BVH_STEPS_QUEUE = 8;   // ... in the level queue's launch, where a finished lane waits for the turn's end with its children (HW11 3.27 / 3.11 / 3.15 / 3.26)

// Builds of the kernels: plain; tallying the tests they execute (crt_options::collect_counters == 2); bounds-checked (crt_tuning::bvh
// == 2: every index is compared with its array's size first; a violation is recorded in the frame's counter block -- word
// SC_BVH_DIAG + 2 code, the index beside it -- and index 0 is read instead: a development build that cannot fault)
enum : int { BVH_PLAIN = 0, BVH_TALLY = 1, BVH_CHECKED = 2 };
constexpr int SC_BVH_DIAG = SC_HEAVY_DIAG + 16;
template <int MODE>
__device__ __forceinline__ uint32_t bvh_at(const KernelArgs &A, const uint32_t i, const uint32_t n, const uint32_t code) {
    if (MODE == BVH_CHECKED && i >= n) { A.f->s_counts[SC_BVH_DIAG + 2 * code] = 1u; A.f->s_counts[SC_BVH_DIAG + 2 * code + 1] = i; return 0u; }
    return i;
}
// The record of filter entry e (48 bytes).  The byte offset is computed in 32 bits ON PURPOSE: the entry index comes out of a masked
// word (`cur & 0x00FFFFFF`), and ROCm 7.2's gfx950 backend lowers zext(x & 0xFFFFFF) * 48 to a v_mad_u64_u32 on the UNMASKED x -- a wild
// address, a memory fault (found in round 3 on another kernel, met again here: the bounds-checked build, whose compare keeps the mask
// alive, never faulted and never recorded a violation).  bvh_build keeps 48 x entries below 2^32.
__device__ __forceinline__ const float4 *bvh_entry(const KernelArgs &A, const uint32_t e) {
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A.s->bvh_tris) + (size_t)(uint32_t)(e * 48u));
}

struct BvhRay { float ix, iy, iz, cpx, cpy, cpz, cmx, cmy, cmz; };

// false: the ray has non-finite coordinates (the filter would pass everything: let the reference-order kernels take it)
__device__ __forceinline__ bool bvh_ray_setup(const KernelArgs &A, const Ray &R, BvhRay &B) {
    const float rho = (A.s->bvh_extent + fmaxf(fmaxf(fabsf(R.ox), fabsf(R.oy)), fabsf(R.oz))) * 0x1p-16f;
    // A direction component of (nearly) zero -- a light straight above its surface point -- would make 1/d infinite and every product
    // with it a NaN, which passes: such a ray would walk half the hierarchy.  The filter walks the ray with that component moved to
    // +-1e-12 instead: over any distance a hit can have (the scene's extent, < 1e7 here or rho is no longer small against it) the two
    // rays are less than 1e-5 rho apart, which the slack covers many times over.
    const float dx = fabsf(R.dx) < 1e-12f ? copysignf(1e-12f, R.dx) : R.dx, dy = fabsf(R.dy) < 1e-12f ? copysignf(1e-12f, R.dy) : R.dy,
                dz = fabsf(R.dz) < 1e-12f ? copysignf(1e-12f, R.dz) : R.dz;
    // (v_rcp_f32: one unit in the last place off the true quotient -- a ray whose axes are scaled by 1 +- 2 u each; the 2^-20 by which both
    //  interval ends are widened covers it many times over, and the exact 1/d of the reference's slab test stays in R for the verification)
    B.ix = __builtin_amdgcn_rcpf(dx); B.iy = __builtin_amdgcn_rcpf(dy); B.iz = __builtin_amdgcn_rcpf(dz);
    B.cpx = -((R.ox + rho) * B.ix); B.cmx = -((R.ox - rho) * B.ix);
    B.cpy = -((R.oy + rho) * B.iy); B.cmy = -((R.oy - rho) * B.iy);
    B.cpz = -((R.oz + rho) * B.iz); B.cmz = -((R.oz - rho) * B.iz);
    const float s = (R.ox + R.oy + R.oz) + (R.dx + R.dy + R.dz);
    return fabsf(s) < INFINITY && fabsf(R.ox) < INFINITY && fabsf(R.oy) < INFINITY && fabsf(R.oz) < INFINITY;  // (NaN: false)
}

// Does the ray come within rho of the box somewhere in [0, tmax]?  May say yes when it does not, never no when it does.
__device__ __forceinline__ bool bvh_child_test(const BvhRay &B, const float lox, const float loy, const float loz, const float hix,
                                               const float hiy, const float hiz, const float tmax, float &tnear) {
    const float ax = __builtin_fmaf(lox, B.ix, B.cpx), bx = __builtin_fmaf(hix, B.ix, B.cmx);
    const float ay = __builtin_fmaf(loy, B.iy, B.cpy), by = __builtin_fmaf(hiy, B.iy, B.cmy);
    const float az = __builtin_fmaf(loz, B.iz, B.cpz), bz = __builtin_fmaf(hiz, B.iz, B.cmz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = tn;
    return !(__builtin_fmaf(-fabsf(tn), BVH_WIDEN, tn) > __builtin_fmaf(fabsf(tf), BVH_WIDEN, tf));
}

// The leaves of `mesh`'s tree that list triangle `tri` (vertices a, b, c), for a triangle listed by too many of them for a list
// (BVH_TRI_WALK): the reference's own walk of the tree (KDTree.cpp:53-74: the threaded nodes, the exact slab test), pruned to the
// nodes whose box overlaps the triangle's -- every leaf that lists it does (checked by bvh_build), and so does every ancestor of
// such a leaf (nested boxes) -- and stopped at the first leaf that passes and lists it: the first in visit order.  A leaf's entries
// ascend (checked by bvh_build) and its compact link says how many there are: the triangle is looked up by bisection.
template <int MODE>
__device__ __forceinline__ bool bvh_leaf_walk(const KernelArgs &A, const Ray &R, const uint32_t tri, const uint32_t mesh, const float4 &a,
                                              const float4 &b, const float4 &c, uint32_t &k3, uint32_t &nbox) {
    const float eps = A.s->bvh_overlap_eps;
    const float tlx = fminf(fminf(a.x, b.x), c.x) - eps, thx = fmaxf(fmaxf(a.x, b.x), c.x) + eps;
    const float tly = fminf(fminf(a.y, b.y), c.y) - eps, thy = fmaxf(fmaxf(a.y, b.y), c.y) + eps;
    const float tlz = fminf(fminf(a.z, b.z), c.z) - eps, thz = fmaxf(fmaxf(a.z, b.z), c.z) + eps;
    uint32_t wn = A.s->meshes[bvh_at<MODE>(A, mesh, A.s->n_meshes, 0)].root;
    uint32_t trips = 0;
    while (wn != END) {   // (links point forward: the walk is finite)
        if (MODE == BVH_CHECKED && ++trips > (1u << 22)) { bvh_at<MODE>(A, trips, 0u, 14); break; }
        wn = bvh_at<MODE>(A, wn, A.s->n_nodes, 1);
        const float4 q0 = A.s->pnodes[2 * (size_t)wn], q1 = A.s->pnodes[2 * (size_t)wn + 1];
        const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
        if (MODE == BVH_TALLY) nbox++;
        const bool overlap = tlx <= q1.x && thx >= q0.x && tly <= q1.y && thy >= q0.y && tlz <= q1.z && thz >= q0.z;
        const bool hit = overlap && slab_test(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
        if (hit && is_leaf_link(link)) {
            uint32_t lo = link & 0x00FFFFFFu, hi = lo + ((link >> 24) & 0x7Fu);  // entries [lo, hi], ascending
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((A.s->leaf_tris[bvh_at<MODE>(A, mid, A.s->n_leaf_tris, 2)] & ~LAST) < tri) lo = mid + 1u; else hi = mid;
            }
            if ((A.s->leaf_tris[bvh_at<MODE>(A, lo, A.s->n_leaf_tris, 3)] & ~LAST) == tri) { k3 = lo; return true; }
            wn = miss;
        } else wn = (hit && link != END) ? link : miss;
    }
    return false;
}

// The first box of the list [j, e) -- 2 x float4 per entry: {lo, entry position} {hi, -} -- the ray passes (exact slab test), two
// entries per round trip; NONE: it passes none of them.
template <int MODE>
__device__ __forceinline__ uint32_t bvh_first_passing(const KernelArgs &A, const Ray &R, const float4 *list, const uint32_t size, uint32_t j,
                                                   uint32_t e, uint32_t &nbox) {
    if (MODE == BVH_CHECKED && (e > size || j > e)) { bvh_at<MODE>(A, e, size, 4); bvh_at<MODE>(A, j, e + 1u, 5); e = j; }
    for (; j < e; j += 2u) {
        const uint32_t j1 = j + 1u < e ? j + 1u : j;
        const float4 p0 = list[2 * (size_t)j], p1 = list[2 * (size_t)j + 1], q0 = list[2 * (size_t)j1], q1 = list[2 * (size_t)j1 + 1];
        if (MODE == BVH_TALLY) nbox += j1 != j ? 2u : 1u;
        if (slab_test(R, p0.x, p0.y, p0.z, p1.x, p1.y, p1.z)) return __float_as_uint(p0.w);
        if (slab_test(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z)) return __float_as_uint(q0.w);   // (j1 == j: the same verdict again)
    }
    return NONE;
}

// Would the reference have tested triangle `tri` (vertices a, b, c) for this ray?  k2 / k3: where it first collects the mesh's / the
// triangle's hit.  (cache_mesh, cache_k2): the last mesh looked up for this ray and its answer (a ray's candidates mostly share a mesh).
template <int MODE>
__device__ __forceinline__ bool bvh_verify(const KernelArgs &A, const Ray &R, const uint32_t tri, const float4 &a, const float4 &b,
                                           const float4 &c, uint32_t &mesh, uint32_t &cache_mesh, uint32_t &cache_k2, uint32_t &k2,
                                           uint32_t &k3, uint32_t &nbox) {
    const uint32_t tm = A.s->tri_mesh[bvh_at<MODE>(A, tri, A.s->n_triangles, 6)];
    mesh = bvh_at<MODE>(A, tm & ~BVH_TRI_WALK, A.s->n_meshes, 7);
    if (mesh != cache_mesh) {
        cache_mesh = mesh;
        cache_k2 = bvh_first_passing<MODE>(A, R, A.s->mesh_top_list, A.s->n_mesh_top_entries, A.s->mesh_top_first[mesh], A.s->mesh_top_first[mesh + 1], nbox);
    }
    k2 = cache_k2;
    if (k2 == NONE) return false;
    if (tm & BVH_TRI_WALK) return bvh_leaf_walk<MODE>(A, R, tri, mesh, a, b, c, k3, nbox);
    k3 = bvh_first_passing<MODE>(A, R, A.s->tri_leaf_list, A.s->n_tri_leaf_entries, A.s->tri_leaf_first[tri], A.s->tri_leaf_first[tri + 1], nbox);
    return k3 != NONE;
}

// the reference's triangle test (kernel_plan.h's expressions; the plane offset is recomputed: -(v0 . n), checked against the stored
// one bit for bit when the filter is built)
__device__ __forceinline__ bool bvh_triangle(const Ray &R, const bool primary, const float4 &a, const float4 &b, const float4 &c, float &t,
                                             float &px, float &py, float &pz) {
    const float nx = a.w, ny = b.w, nz = c.w;
    const float nd = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
    const float plane = -dot3(a.x, a.y, a.z, nx, ny, nz);  // distanceToPlane, Ray.cpp:17
    t = -(dot3(nx, ny, nz, R.ox, R.oy, R.oz) + plane) / nd;
    px = R.ox + R.dx * t; py = R.oy + R.dy * t; pz = R.oz + R.dz * t;
    float s0, s1, s2;
    {
        const float ex = b.x - a.x, ey = b.y - a.y, ez = b.z - a.z, cx = px - a.x, cy = py - a.y, cz = pz - a.z;
        s0 = dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx);
    }
    {
        const float ex = c.x - b.x, ey = c.y - b.y, ez = c.z - b.z, cx = px - b.x, cy = py - b.y, cz = pz - b.z;
        s1 = dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx);
    }
    {
        const float ex = a.x - c.x, ey = a.y - c.y, ez = a.z - c.z, cx = px - c.x, cy = py - c.y, cz = pz - c.z;
        s2 = dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx);
    }
    return !(primary && nd >= 0) && !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
}

// What a lane carries through a walk.  `cur`: the node or (rest of a) leaf to look at next, BVH_EMPTY: take one off the stack.
struct BvhWalk {
    BvhRay B;
    uint32_t cur, sp;
    float best;                 // closest hit: distance of the best verified hit so far; shadow: the end of the segment
    uint32_t bk2, bk3, btri, bmesh, cache_mesh, cache_k2;
    bool have, give_up;
    uint32_t steps;             // BVH_TALLY: steps of this walk (diagnostics)
};
__device__ __forceinline__ void bvh_walk_begin(BvhWalk &W, const float tmax) {
    W.cur = 0; W.sp = 0;  // the root is node 0
    W.best = tmax;
    W.bk2 = W.bk3 = NONE; W.btri = 0; W.bmesh = 0; W.cache_mesh = NONE; W.cache_k2 = NONE;
    W.have = false; W.give_up = false; W.steps = 0;
}

// One step of a lane's walk: an inner node (its four children tested, the nearest taken, the others pushed) or two triangles of a
// leaf.  false: the walk is over -- nothing left, the stack too small (give_up), or (SHADOW) an occluder found (have).
// KIND 0: a closest-hit walk; 1: a shadow walk; 2: whichever `shadow_lane` says, lane by lane -- the level queue's waves hold both kinds
// at once, and one instruction stream for both halves what a turn costs them (the kinds differ in what a leaf does with an accepted hit).
template <int KIND, int MODE>
__device__ __forceinline__ bool bvh_step(const KernelArgs &A, const Ray &R, const bool primary, const float light_dist, const bool every_mesh,
                                         BvhWalk &W, const BvhStack &stack, uint32_t &nbox, uint32_t &ntri, const bool shadow_lane = false) {
    const bool SHADOW = KIND == 1 || (KIND == 2 && shadow_lane);
    if (MODE == BVH_TALLY) W.steps++;
    if (W.cur == BVH_EMPTY) {
        if (W.sp == 0) return false;
        bvh_at<MODE>(A, W.sp - 1u, A.s->bvh_stack, 8);
        W.cur = bvh_pop(stack, W.sp);
    }
    if (!(W.cur & BVH_LEAF)) {
        const float4 *N = A.s->bvh_nodes + 8 * (size_t)bvh_at<MODE>(A, W.cur, A.s->n_bvh_nodes, 9);
        const float4 lx = N[0], ly = N[1], lz = N[2], hx = N[3], hy = N[4], hz = N[5];
        const float4 ch = N[6];
        if (MODE == BVH_TALLY) nbox += 4;
        uint32_t c0 = __float_as_uint(ch.x), c1 = __float_as_uint(ch.y), c2 = __float_as_uint(ch.z), c3 = __float_as_uint(ch.w);
        float t0, t1, t2, t3;
        const bool h0 = bvh_child_test(W.B, lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, W.best, t0) && c0 != BVH_EMPTY;
        const bool h1 = bvh_child_test(W.B, lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, W.best, t1) && c1 != BVH_EMPTY;
        const bool h2 = bvh_child_test(W.B, lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, W.best, t2) && c2 != BVH_EMPTY;
        const bool h3 = bvh_child_test(W.B, lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, W.best, t3) && c3 != BVH_EMPTY;
        // nearest first: sort the four (entry distance, child) pairs, misses last (a NaN distance sorts anywhere: order is only speed)
        t0 = h0 ? t0 : INFINITY; t1 = h1 ? t1 : INFINITY; t2 = h2 ? t2 : INFINITY; t3 = h3 ? t3 : INFINITY;
        c0 = h0 ? c0 : BVH_EMPTY; c1 = h1 ? c1 : BVH_EMPTY; c2 = h2 ? c2 : BVH_EMPTY; c3 = h3 ? c3 : BVH_EMPTY;
#define BVH_CSWAP(ta, ca, tb, cb) { const bool sw = tb < ta; const float tt = sw ? tb : ta; tb = sw ? ta : tb; ta = tt; const uint32_t cc = sw ? cb : ca; cb = sw ? ca : cb; ca = cc; }
        BVH_CSWAP(t0, c0, t1, c1) BVH_CSWAP(t2, c2, t3, c3) BVH_CSWAP(t0, c0, t2, c2) BVH_CSWAP(t1, c1, t3, c3) BVH_CSWAP(t1, c1, t2, c2)
#undef BVH_CSWAP
        const uint32_t n_push = (c1 != BVH_EMPTY ? 1u : 0u) + (c2 != BVH_EMPTY ? 1u : 0u) + (c3 != BVH_EMPTY ? 1u : 0u);
        if (W.sp + n_push > A.s->bvh_stack) { W.give_up = true; return false; }   // (cannot happen: the stack is sized by the depth)
        if (c3 != BVH_EMPTY) bvh_push(stack, W.sp, c3);
        if (c2 != BVH_EMPTY) bvh_push(stack, W.sp, c2);
        if (c1 != BVH_EMPTY) bvh_push(stack, W.sp, c1);
        W.cur = c0;  // (BVH_EMPTY when the nearest slot holds a miss -- nothing passed, or a NaN distance out of order: the next step pops)
    } else {
        // up to TPS triangles of the leaf per step, all fetched before any is tested (one round trip); what is left of the leaf stays in `cur`
        constexpr uint32_t TPS = 2u;   // (four per step for the closest-hit walks: measured, no faster, 30 registers more)
        const uint32_t left = (W.cur >> 24) & 0x7Fu;  // `left` more after the first
        const uint32_t n_here = left + 1u < TPS ? left + 1u : TPS;
        const uint32_t first = bvh_at<MODE>(A, W.cur & 0x00FFFFFFu, A.s->n_bvh_entries - (n_here - 1u), 10);
        float4 ta[TPS], tb[TPS], tc[TPS];
        uint32_t tid[TPS];
#pragma unroll
        for (uint32_t k = 0; k < TPS; k++) {
            const uint32_t e = first + (k < n_here ? k : n_here - 1u);   // (beyond the leaf: its last triangle again, not tested)
            const float4 *T = bvh_entry(A, e);
            ta[k] = T[0]; tb[k] = T[1]; tc[k] = T[2];
            tid[k] = A.s->bvh_ids[e];
        }
        W.cur = left >= TPS ? (BVH_LEAF | ((left - TPS) << 24) | (first + TPS)) : BVH_EMPTY;
#pragma unroll
        for (uint32_t k = 0; k < TPS; k++) {
            if (k >= n_here) break;
            const float4 &a = ta[k], &b = tb[k], &c = tc[k];
            const uint32_t id = tid[k];
            if (SHADOW && (id & BVH_ID_REFRACTIVE) && !every_mesh) continue;  // AccelerationStructure.cpp:66-71 (not in the GI mode)
            float t, px, py, pz;
            if (MODE == BVH_TALLY) ntri++;
            const bool ok = bvh_triangle(R, primary, a, b, c, t, px, py, pz);
            if (!(ok && t < INFINITY)) continue;
            uint32_t mesh, k2, k3 = NONE;
            if (SHADOW) {
                if (shadow_hit_occludes(R, px, py, pz, light_dist) &&
                    bvh_verify<MODE>(A, R, id & ~BVH_ID_REFRACTIVE, a, b, c, mesh, W.cache_mesh, W.cache_k2, k2, k3, nbox)) { W.have = true; return false; }
            } else if (t <= W.best) {
                const uint32_t tri = id & ~BVH_ID_REFRACTIVE;
                if (bvh_verify<MODE>(A, R, tri, a, b, c, mesh, W.cache_mesh, W.cache_k2, k2, k3, nbox) &&
                    (t < W.best || k2 < W.bk2 || (k2 == W.bk2 && k3 < W.bk3))) {
                    W.best = t; W.bk2 = k2; W.bk3 = k3; W.btri = tri; W.bmesh = mesh; W.have = true;
                }
            }
        }
    }
    return true;
}

// a lane: wants a ray; walks it through the filter; nothing left to fetch; its walk has ended; checks the miss its walk ended with
enum : int { BVH_FETCH = 0, BVH_WALK = 1, BVH_OUT = 2, BVH_FINISHED = 3, BVH_MISS_CHECK = 4,
             BVH_WAIT = 5,       // bvh_trace_queue: for the record of the ray it has claimed
             BVH_SHADOWS = 6 };  // bvh_trace_queue: walks the shadow rays of the diffuse hit it has just shaded

// The miss check's view of the ray: its whole LINE (the reference's slab test has no t >= 0, BoundingBox.h:85-108), with the reference's
// notion of a parallel axis -- |d| < FLT_EPSILON is a containment test there (BoundingBox.h:90-93) -- kept as a superset: such a
// component becomes +-1e-30, whose slab interval holds every moderate t exactly when the origin lies in the (slackened) slab.
__device__ __forceinline__ void bvh_line_setup(const KernelArgs &A, const Ray &R, BvhRay &B) {
    const float rho = (A.s->bvh_extent + fmaxf(fmaxf(fabsf(R.ox), fabsf(R.oy)), fabsf(R.oz))) * 0x1p-16f;
    const float dx = fabsf(R.dx) < 2.0f * FLT_EPSILON ? copysignf(1e-30f, R.dx) : R.dx, dy = fabsf(R.dy) < 2.0f * FLT_EPSILON ? copysignf(1e-30f, R.dy) : R.dy,
                dz = fabsf(R.dz) < 2.0f * FLT_EPSILON ? copysignf(1e-30f, R.dz) : R.dz;
    B.ix = 1.0f / dx; B.iy = 1.0f / dy; B.iz = 1.0f / dz;
    B.cpx = -((R.ox + rho) * B.ix); B.cmx = -((R.ox - rho) * B.ix);
    B.cpy = -((R.oy + rho) * B.iy); B.cmy = -((R.oy - rho) * B.iy);
    B.cpz = -((R.oz + rho) * B.iz); B.cmz = -((R.oz - rho) * B.iz);
}
// Does the line come within rho of the box?  (May say yes when it does not, never no when the reference's slab test passes.)
__device__ __forceinline__ bool bvh_line_test(const BvhRay &B, const float lox, const float loy, const float loz, const float hix,
                                              const float hiy, const float hiz) {
    const float ax = __builtin_fmaf(lox, B.ix, B.cpx), bx = __builtin_fmaf(hix, B.ix, B.cmx);
    const float ay = __builtin_fmaf(loy, B.iy, B.cpy), by = __builtin_fmaf(hiy, B.iy, B.cmy);
    const float az = __builtin_fmaf(loz, B.iz, B.cpz), bz = __builtin_fmaf(hiz, B.iz, B.cmz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return !(__builtin_fmaf(-fabsf(tn), BVH_WIDEN, tn) > __builtin_fmaf(fabsf(tf), BVH_WIDEN, tf));
}

// One step of the miss check (W.B holds the line): an inner node -- the children whose reach box the line passes and whose normal
// cone comes within 1e-4 rad of perpendicular to d are kept -- or two triangles of a leaf: one that the reference's test accepts,
// whatever its distance, and that the reference would have tested (bvh_verify) refutes the miss: W.give_up.  false: the check is over.
template <int MODE>
__device__ __forceinline__ bool bvh_miss_step(const KernelArgs &A, const Ray &R, const bool primary, BvhWalk &W, const BvhStack &stack, uint32_t &nbox,
                                              uint32_t &ntri) {
    if (MODE == BVH_TALLY) W.steps++;
    if (W.cur == BVH_EMPTY) {
        if (W.sp == 0) return false;
        bvh_at<MODE>(A, W.sp - 1u, A.s->bvh_stack, 8);
        W.cur = bvh_pop(stack, W.sp);
    }
    if (!(W.cur & BVH_LEAF)) {
        const uint32_t node = bvh_at<MODE>(A, W.cur, A.s->n_bvh_nodes, 9);
        const float4 *V = A.s->bvh_vnodes + 6 * (size_t)node, *C = A.s->bvh_cones + 4 * (size_t)node;
        const float4 lx = V[0], ly = V[1], lz = V[2], hx = V[3], hy = V[4], hz = V[5];
        const float4 cx = C[0], cy = C[1], cz = C[2], ck = C[3];
        const float4 ch = A.s->bvh_nodes[8 * (size_t)node + 6];
        if (MODE == BVH_TALLY) nbox += 4;
        const uint32_t c0 = __float_as_uint(ch.x), c1 = __float_as_uint(ch.y), c2 = __float_as_uint(ch.z), c3 = __float_as_uint(ch.w);
        const bool h0 = c0 != BVH_EMPTY && !(fabsf(__builtin_fmaf(cx.x, R.dx, __builtin_fmaf(cy.x, R.dy, cz.x * R.dz))) > ck.x) && bvh_line_test(W.B, lx.x, ly.x, lz.x, hx.x, hy.x, hz.x);
        const bool h1 = c1 != BVH_EMPTY && !(fabsf(__builtin_fmaf(cx.y, R.dx, __builtin_fmaf(cy.y, R.dy, cz.y * R.dz))) > ck.y) && bvh_line_test(W.B, lx.y, ly.y, lz.y, hx.y, hy.y, hz.y);
        const bool h2 = c2 != BVH_EMPTY && !(fabsf(__builtin_fmaf(cx.z, R.dx, __builtin_fmaf(cy.z, R.dy, cz.z * R.dz))) > ck.z) && bvh_line_test(W.B, lx.z, ly.z, lz.z, hx.z, hy.z, hz.z);
        const bool h3 = c3 != BVH_EMPTY && !(fabsf(__builtin_fmaf(cx.w, R.dx, __builtin_fmaf(cy.w, R.dy, cz.w * R.dz))) > ck.w) && bvh_line_test(W.B, lx.w, ly.w, lz.w, hx.w, hy.w, hz.w);
        const uint32_t n_hit = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
        if (W.sp + n_hit > A.s->bvh_stack) { W.give_up = true; return false; }   // (cannot happen: the stack is sized by the depth)
        W.cur = BVH_EMPTY;
        if (h0) { W.cur = c0; }
        if (h1) { if (W.cur != BVH_EMPTY) bvh_push(stack, W.sp, W.cur); W.cur = c1; }
        if (h2) { if (W.cur != BVH_EMPTY) bvh_push(stack, W.sp, W.cur); W.cur = c2; }
        if (h3) { if (W.cur != BVH_EMPTY) bvh_push(stack, W.sp, W.cur); W.cur = c3; }
    } else {
        const uint32_t left = (W.cur >> 24) & 0x7Fu;
        const uint32_t first = bvh_at<MODE>(A, W.cur & 0x00FFFFFFu, A.s->n_bvh_entries - (left ? 1u : 0u), 10);
        const uint32_t second = first + (left ? 1u : 0u);
        const float4 *T0 = bvh_entry(A, first), *T1 = bvh_entry(A, second);
        const float4 a0 = T0[0], b0 = T0[1], c0 = T0[2], a1 = T1[0], b1 = T1[1], c1 = T1[2];
        const uint32_t id0 = A.s->bvh_ids[first], id1 = A.s->bvh_ids[second];
        W.cur = left >= 2u ? (BVH_LEAF | ((left - 2u) << 24) | (first + 2u)) : BVH_EMPTY;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !left) break;
            const float4 &a = k ? a1 : a0, &b = k ? b1 : b0, &c = k ? c1 : c0;
            float t, px, py, pz;
            if (MODE == BVH_TALLY) ntri++;
            if (!bvh_triangle(R, primary, a, b, c, t, px, py, pz)) continue;
            uint32_t mesh, k2, k3 = NONE;
            if (bvh_verify<MODE>(A, R, (k ? id1 : id0) & ~BVH_ID_REFRACTIVE, a, b, c, mesh, W.cache_mesh, W.cache_k2, k2, k3, nbox)) { W.give_up = true; return false; }
        }
    }
    return true;
}

// Every ray of recursion level `gen`, persistent waves: a lane walks its ray through the filter; a ray that ends WITHOUT a verified
// finite hit goes through the miss check (above) in the same lane.  When enough lanes of the wave have finished, they shade together
// -- shootRay's material dispatch (kernel_stream.h: shade_and_emit) -- and take the next rays of the level (consecutive ones: at
// level 0 neighbouring pixels of a tile).
template <int MODE, bool LQ = false, bool GI = false>   // LQ: level 0 of a frame whose deeper levels are bvh_trace_queue's: the children go to the level queue; GI: the GI / multi-sample mode's build
__device__ __forceinline__ void bvh_shade_level(const KernelArgs &A, const uint32_t gen, uint32_t *stack_lds) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const float4 *in_q = A.f->s_rayq[gen & 1u];
    const BvhStack stack = bvh_stack_of(A, stack_lds, false);
    const bool primary = gen == 0;
    uint32_t nbox = 0, ntri = 0;
    Ray R;
    R.ox = R.oy = R.oz = R.dx = R.dy = R.dz = R.ix = R.iy = R.iz = 0; R.parmask = 0;
    BvhWalk W;
    bvh_walk_begin(W, INFINITY);
    int state = BVH_FETCH;
    uint32_t r = 0, spins = 0;
    WaveChunk chunk{0u, 0u};
    for (;;) {
        if (MODE == BVH_CHECKED && ++spins > (1u << 22)) { bvh_at<MODE>(A, spins, 0u, 15); break; }   // (a loop that does not end: say so and leave)
        const uint32_t n_free = (uint32_t)__popcll(__ballot(state == BVH_FETCH || state == BVH_FINISHED));
        if (n_free >= BVH_BATCH || (n_free && !__ballot(state == BVH_WALK || state == BVH_MISS_CHECK))) {
            if (state == BVH_FINISHED) {
                if (MODE == BVH_TALLY && gen < 10u) {   // diagnostics per level: the longest walk, all steps / 16, walks
                    uint32_t *dg = A.f->s_counts + SC_BVH_DIAG + 48 + 3 * gen;
                    atomicMax(dg + 0, W.steps); atomicAdd(dg + 1, W.steps); atomicAdd(dg + 2, 1u);
                }
                if (W.give_up) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; }   // a ray the filter cannot take, a miss refuted: the frame goes to the last resort
                // (the GI build only walks: the hit goes to the level's record and stream_shade_all<true> shades the level in a launch of its
                //  own with every lane busy -- with the sample directions and the gi_samples child rays inlined here the kernel needs 231 VGPRs)
                if constexpr (GI) A.f->s_hits_all[r] = make_float4(W.best, __uint_as_float(W.btri), __uint_as_float(W.bmesh), __uint_as_float(W.have ? 1u : 0u));
                else shade_and_emit<false, false, LQ>(A, gen, r, node_base, child_base, R, W.have, W.best, W.btri, W.bmesh, nullptr, lane);
                state = BVH_FETCH;
            }
            // one fetch per free lane and round (a lane whose ray needs no walk here -- an uncovered pixel -- asks again next round): no
            // inner loop, no `continue` (compiler notes)
            // (level 0: millions of rays, claimed A.chunk at a time; a deeper level's few thousand are dealt out one refill at a time)
            uint32_t claimed = 0;
            if (gen == 0) claimed = wave_fetch_chunked(A.f->s_counts + SC_FETCH, lane, state == BVH_FETCH, chunk, A.chunk, count);
            if (state == BVH_FETCH) {
                r = gen == 0 ? claimed : wave_fetch(A.f->s_counts + SC_FETCH + gen, lane);
                if (r >= count) state = BVH_OUT;
                else {
                    bool walk = true;
                    if (gen == 0) {
                        const Level0Ray P = level0_decode<GI>(A, r);
                        if (!P.covered) {
                            reinterpret_cast<uint32_t *>(A.f->s_nodes + 2 * (size_t)r)[3] = TN_SKIP;
                            level0_release_shadow_slots(A, r);
                            if constexpr (GI) A.f->s_hits_all[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(3u));   // (not a ray: stream_shade_all skips it)
                            walk = false;
                        } else level0_ray<GI>(A, P, R);
                    } else {
                        const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
                        R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                        R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
                        normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                        ray_prepare(R);
                    }
                    if (walk) {
                        bvh_walk_begin(W, INFINITY);
                        if (bvh_ray_setup(A, R, W.B)) state = BVH_WALK;
                        else { W.give_up = true; state = BVH_FINISHED; }
                    }
                }
            }
        }
        if (!__ballot(state != BVH_OUT)) break;
        if (state == BVH_WALK) {
            for (int it = 0; it < BVH_STEPS; ++it)
                if (state == BVH_WALK && !bvh_step<0, MODE>(A, R, primary, 0.0f, false, W, stack, nbox, ntri)) {
                    if (W.have || W.give_up) state = BVH_FINISHED;
                    else { W.cur = 0; W.sp = 0; W.cache_mesh = NONE; W.cache_k2 = NONE; bvh_line_setup(A, R, W.B); state = BVH_MISS_CHECK; }
                }
        }
        if (state == BVH_MISS_CHECK) {
            for (int it = 0; it < BVH_STEPS; ++it)
                if (state == BVH_MISS_CHECK && !bvh_miss_step<MODE>(A, R, primary, W, stack, nbox, ntri)) state = BVH_FINISHED;
        }
    }
    exec_counters_flush(A, nbox, ntri, lane);
}
__global__ __launch_bounds__(BLOCK) void bvh_trace_shade(const KernelArgs A, const uint32_t gen) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_shade_level<BVH_PLAIN>(A, gen, stack_lds);
}
// the GI / multi-sample mode's levels (jittered samples of a pixel at level 0, gi_samples child rays per diffuse hit: kernel_stream.h,
// shade_and_emit<., true>) through the same filter: only the rays' origin and the shading differ, and the shadow rays skip no mesh
__global__ __launch_bounds__(BLOCK) void bvh_trace_shade_gi(const KernelArgs A, const uint32_t gen) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_shade_level<BVH_PLAIN, false, true>(A, gen, stack_lds);
}
__global__ __launch_bounds__(BLOCK) void bvh_trace_shade_tally(const KernelArgs A, const uint32_t gen) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_shade_level<BVH_TALLY>(A, gen, stack_lds);
}
__global__ __launch_bounds__(BLOCK) void bvh_trace_shade_checked(const KernelArgs A, const uint32_t gen) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_shade_level<BVH_CHECKED>(A, gen, stack_lds);
}
template <int MODE>
__global__ __launch_bounds__(BLOCK) void bvh_trace_level0(const KernelArgs A) {   // (bounded to 128 VGPRs for four waves per SIMD: 2 spills, 3.40 vs 3.31 ms)
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_shade_level<MODE, true>(A, 0u, stack_lds);
    // The level queue's launch runs BESIDE this one and may only end once nothing but its own rays can reserve entries any more: the
    // last workgroup to get here says so (LQ_LEVEL0).  Every reservation of a workgroup has RETURNED (an atomic with a result) before
    // the workgroup counts itself; the flag is the launch's own word, not a kernel behind it: a kernel behind it may be held back until
    // the queue's launch has ended (a profiler that serialises launches does that), and the queue's launch would wait for it for ever.
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(A.f->s_lq_words + LQ_LEVEL0 + 1, 1u) == gridDim.x - 1u)
        __hip_atomic_store((uint32_t __attribute__((address_space(1))) *)(A.f->s_lq_words + LQ_LEVEL0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// EVERY ray below level 0, in one launch (crt_tuning::level_queue).  Launched level by level a frame waits, at each of its levels,
// for that level's longest walk -- eight times 0.2 - 0.3 ms on the benchmark frame for walks that average a tenth of it.  Here a lane
// takes the next ray of the level queue (kernel_stream.h: lq_store_ray), walks it, shades it, appends its children to the same queue
// and takes the next one: a pixel's chain of rays never waits for another pixel's.
//   rays are RESERVED (SC_LQ_TAIL, when their parent is shaded), CLAIMED (SC_LQ_HEAD, by the lane that will walk them) and counted
//   DONE (SC_LQ_DONE, after their own children are reserved); entry k owns ray-tree node count0 + k.
//   A lane that has claimed an entry polls its record until every granule carries the frame's tag -- the record may be written by a
//   workgroup on another XCD while this one runs: agent-scope stores and loads, no fence (Guideline 16, R2).
//   The end: done == tail, read in that order (done never exceeds tail and both only grow: equal values read that way were equal at
//   the second read, and once equal nothing is in flight that could reserve more).  Every wave reaches it: a wave only waits for
//   records of rays reserved by waves that are running (a reserved ray's parent was walked by a resident wave, which writes the
//   record before it does anything else), the overflow word ends every wait, and a loop that turns 2^20 times raises it.
template <int MODE>
__device__ __forceinline__ void bvh_queue_levels(const KernelArgs &A, uint32_t *stack_lds) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    typedef uint32_t __attribute__((address_space(1))) *gu32;
    const gu32 tail_p = (gu32)(A.f->s_lq_words + LQ_TAIL), head_p = (gu32)(A.f->s_lq_words + LQ_HEAD), done_p = (gu32)(A.f->s_lq_words + LQ_DONE),
               ovf_p = (gu32)(A.f->s_lq_words + LQ_ABORT), l0_p = (gu32)(A.f->s_lq_words + LQ_LEVEL0);
    const uint32_t count0 = stream_level_count(A, 0);
    const BvhStack stack = bvh_stack_of(A, stack_lds, false);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t nbox = 0, ntri = 0;
    Ray R;
    R.ox = R.oy = R.oz = R.dx = R.dy = R.dz = R.ix = R.iy = R.iz = 0; R.parmask = 0;
    BvhWalk W;
    bvh_walk_begin(W, INFINITY);
    int state = BVH_FETCH;
    uint32_t r = 0, gen = 1, spins = 0, idle_polls = 0, early_polls = 0;
    // the shadow rays of a diffuse hit are walked by the lane that shaded it, one light after the other (they have no children: nothing to
    // hand over, and no launch of their own behind the queue): first slot and stride in the shadow queue, light in turn, walk under way?
    uint32_t sh_first = 0, sh_stride = 0, sh_li = 0, sh_slot = 0;
    bool sh_walking = false;
    float light_dist = 0;
    const uint32_t n_lights = A.s->n_lights;
    for (;;) {
        if (++spins > (1u << 20)) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; __hip_atomic_store(ovf_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        const unsigned long long fin = __ballot(state == BVH_FINISHED), idle = __ballot(state == BVH_FETCH);
        const bool walking = __ballot(state == BVH_WALK || state == BVH_MISS_CHECK || state == BVH_SHADOWS) != 0;
        // a round of housekeeping -- shading, counting, claiming -- costs the lanes that walk two round trips to the queue's words: when
        // enough walks have finished, when nothing (or little) walks, and otherwise, for the idle lanes' sake, every sixteenth turn
        const uint32_t n_walk = (uint32_t)__popcll(__ballot(state == BVH_WALK || state == BVH_MISS_CHECK || state == BVH_SHADOWS));
        if ((uint32_t)__popcll(fin) >= BVH_BATCH || !walking || (fin && n_walk <= 8u) || ((fin | idle) && (spins & A.bundle) == 0u)) {
            if (state == BVH_FINISHED) {
                if (MODE == BVH_TALLY && gen < 10u) {
                    uint32_t *dg = A.f->s_counts + SC_BVH_DIAG + 48 + 3 * gen;
                    atomicMax(dg + 0, W.steps); atomicAdd(dg + 1, W.steps); atomicAdd(dg + 2, 1u);
                }
                if (W.give_up) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; __hip_atomic_store(ovf_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                bool diffuse = false;
                float keep[6];
                uint32_t keep_index = NONE;
                shade_and_emit<false, false, true>(A, gen, r, count0, count0, R, W.have, W.best, W.btri, W.bmesh, nullptr, lane, &diffuse, &sh_first, &sh_stride,
                                                   A.force_whole ? nullptr : keep, &keep_index);   // (force_whole: every child through the queue, crt_tuning::level_queue | 256)
                state = (diffuse && n_lights) ? BVH_SHADOWS : BVH_FETCH;
                sh_li = 0; sh_walking = false;
                if (keep_index != NONE) {
                    // the reflection child goes on in this lane: no hand-over through memory for the chain it heads (its entry is marked taken)
                    r = keep_index; gen = gen + 1u;
                    R.ox = keep[0]; R.oy = keep[1]; R.oz = keep[2]; R.dx = keep[3]; R.dy = keep[4]; R.dz = keep[5];
                    normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                    ray_prepare(R);
                    bvh_walk_begin(W, INFINITY);
                    if (bvh_ray_setup(A, R, W.B)) state = BVH_WALK;
                    else { W.give_up = true; state = BVH_FINISHED; }
                }
            }
            // (the children's reservations above have returned: the finished rays may count as done)
            uint32_t t = 0, h = 0;
            if (lane == 0) {
                if (fin) atomicAdd(A.f->s_lq_words + LQ_DONE, (uint32_t)__popcll(fin));
                t = __hip_atomic_load(tail_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h = __hip_atomic_load(head_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            t = __builtin_amdgcn_readfirstlane(t); h = __builtin_amdgcn_readfirstlane(h);
            const unsigned long long want = __ballot(state == BVH_FETCH);
            const uint32_t n_want = (uint32_t)__popcll(want), avail = t > h ? t - h : 0u;
            const uint32_t take = n_want < avail ? n_want : avail;
            if (take) {   // (wave-uniform)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(A.f->s_lq_words + LQ_HEAD, take);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t rank = (uint32_t)__popcll(want & below);
                if (state == BVH_FETCH && rank < take) { r = base + rank; state = BVH_WAIT; }
                idle_polls = 0;
            } else if (n_want || __ballot(state == BVH_WAIT)) {
                // nothing to claim: is it the end?
                uint32_t d = 0, t2 = 0, ovf = 0, l0 = 0;
                if (lane == 0) {
                    // (the launch starts WITH level 0, whose rays reserve entries too: has it ended?  Read first: its reservations are in
                    //  `tail` by then)
                    l0 = __hip_atomic_load(l0_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    d = __hip_atomic_load(done_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // done BEFORE tail (see above)
                    t2 = __hip_atomic_load(tail_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ovf = __hip_atomic_load(ovf_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                d = __builtin_amdgcn_readfirstlane(d); t2 = __builtin_amdgcn_readfirstlane(t2); ovf = __builtin_amdgcn_readfirstlane(ovf);
                l0 = __builtin_amdgcn_readfirstlane(l0);
                // Before level 0 has ended an empty queue is no reason to leave -- up to a point: nothing guarantees that level 0 RUNS beside
                // this launch (two streams may share a hardware queue), so the patience is bounded, and a second launch behind level 0
                // (crt_launch.hip) takes whatever this one left.
                if (!l0 && early_polls < 160u) { early_polls++; idle_polls = 0; }
                // nothing is in flight and nothing can be reserved any more (or the frame is lost): whoever has no ray leaves
                if ((ovf || (l0 && d == t2)) && (state == BVH_FETCH || ovf || (state == BVH_WAIT && r >= t2))) state = BVH_OUT;
                // A wave without a single ray that keeps finding nothing to claim leaves too: the waves that hold the rays in flight
                // claim what those rays reserve (a wave that has just reserved rays looks for work at once), and a frame's levels shrink:
                // most waves are only needed for the first of them -- and every idle wave is one more poller of the queue's words.
                if (!__ballot(state != BVH_FETCH && state != BVH_OUT) && ++idle_polls >= 4u) state = BVH_OUT;
            }
        }
        if (state == BVH_WAIT) {   // is my ray there?
            uint32_t level = 0;
            const int what = lq_load_ray(A, r, R.ox, R.oy, R.oz, R.dx, R.dy, R.dz, level);
            if (what == LQ_TAKEN) state = BVH_FETCH;   // its parent's lane walks it: nothing to do (and nothing to count) here
            if (what == LQ_RAY) {
                gen = level;
                normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                ray_prepare(R);
                bvh_walk_begin(W, INFINITY);
                if (bvh_ray_setup(A, R, W.B)) state = BVH_WALK;
                else { W.give_up = true; state = BVH_FINISHED; }
            }
        }
        if (!__ballot(state != BVH_OUT)) break;
        if (state == BVH_SHADOWS && !sh_walking) {   // the next light's ray, as shade_hit queued it
            sh_slot = sh_first + sh_li * sh_stride;
            const float4 q0 = A.f->s_shadowq[2 * (size_t)sh_slot], q1 = A.f->s_shadowq[2 * (size_t)sh_slot + 1];
            if (q1.w == 0.0f) {   // a light behind the surface: no walk (kernel_plan.h has the argument)
                A.f->s_occluded[sh_slot] = 0;
                if (++sh_li == n_lights) state = BVH_FETCH;
            } else {
                R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
                ray_prepare(R);
                light_dist = q0.w;
                bvh_walk_begin(W, light_dist * (1.0f + 0x1p-16f));
                if (bvh_ray_setup(A, R, W.B)) sh_walking = true;
                else { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; A.f->s_occluded[sh_slot] = 0; if (++sh_li == n_lights) state = BVH_FETCH; }
            }
        }
        if (!__ballot(state == BVH_WALK || state == BVH_MISS_CHECK || state == BVH_FINISHED || state == BVH_SHADOWS)) {   // only waiting: poll gently, and ever more gently
            __builtin_amdgcn_s_sleep(127);
            if (idle_polls >= 1u) __builtin_amdgcn_s_sleep(127);
            if (idle_polls >= 2u) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
        }
        // closest-hit walks and shadow walks step TOGETHER, through one instruction stream (bvh_step<2>): a wave here holds both kinds most of
        // the time -- a diffuse hit is followed by its lights' walks in the same lane -- and two streams were two passes over every turn
        // (HW14 2.89 -> 2.76 ms, HW11 3.83 -> 3.48)
        {
            const bool sh_lane = state == BVH_SHADOWS;
            bool stepping = state == BVH_WALK || (sh_lane && sh_walking);
            for (int it = 0; it < BVH_STEPS_QUEUE; ++it)
                if (stepping && !bvh_step<2, MODE>(A, R, false, light_dist, false, W, stack, nbox, ntri, sh_lane)) {
                    stepping = false;
                    if (sh_lane) {
                        if (W.give_up) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; }
                        A.f->s_occluded[sh_slot] = W.have ? 1 : 0;
                        sh_walking = false;
                        if (++sh_li == n_lights) state = BVH_FETCH;
                    } else if (W.have || W.give_up) state = BVH_FINISHED;
                    else { W.cur = 0; W.sp = 0; W.cache_mesh = NONE; W.cache_k2 = NONE; bvh_line_setup(A, R, W.B); state = BVH_MISS_CHECK; }
                }
        }
        if (state == BVH_MISS_CHECK) {
            for (int it = 0; it < BVH_STEPS_QUEUE; ++it)
                if (state == BVH_MISS_CHECK && !bvh_miss_step<MODE>(A, R, false, W, stack, nbox, ntri)) state = BVH_FINISHED;
        }
    }
    exec_counters_flush(A, nbox, ntri, lane);
}
template <int MODE>
__global__ __launch_bounds__(BLOCK) void bvh_trace_queue(const KernelArgs A) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    bvh_queue_levels<MODE>(A, stack_lds);
}

// The shadow rays [first, first + total) of the queue (pass 0: level 0's fixed slots; pass 1: the deeper levels'); `cursor` hands them
// out, consecutive slots to the lanes that refill together: one tile's rays towards one light (kernel_stream.h: level0_shadow_place).
template <int MODE, bool CHUNKED>
__device__ __forceinline__ void bvh_shadow_rays(const KernelArgs &A, const uint32_t first, const uint32_t total, uint32_t *cursor, uint32_t *stack_lds) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const BvhStack stack = bvh_stack_of(A, stack_lds, true);   // (the bulk pass runs beside the level kernels: a region of its own)
    const bool every_mesh = A.f->use_gi != 0;
    uint32_t nbox = 0, ntri = 0;
    Ray R;
    R.ox = R.oy = R.oz = R.dx = R.dy = R.dz = R.ix = R.iy = R.iz = 0; R.parmask = 0;
    BvhWalk W;
    bvh_walk_begin(W, INFINITY);
    float light_dist = 0;
    int state = BVH_FETCH;
    uint32_t r = 0, spins = 0;
    WaveChunk chunk{0u, 0u};
    for (;;) {
        if (MODE == BVH_CHECKED && ++spins > (1u << 22)) { bvh_at<MODE>(A, spins, 0u, 15); break; }   // (a loop that does not end: say so and leave)
        const uint32_t n_free = (uint32_t)__popcll(__ballot(state == BVH_FETCH || state == BVH_FINISHED));
        if (n_free >= BVH_BATCH || (n_free && !__ballot(state == BVH_WALK))) {
            if (state == BVH_FINISHED) {
                if (MODE == BVH_TALLY) {   // diagnostics: the longest walk of the pass, which ray it was, all steps, walks of 1024 steps and more
                    uint32_t *dg = A.f->s_counts + SC_BVH_DIAG + 40;
                    if (atomicMax(dg + 0, W.steps) < W.steps) dg[1] = r;
                    atomicAdd(dg + 2, W.steps >> 4);
                    if (W.steps >= 1024u) atomicAdd(dg + 3, 1u);
                }
                if (W.give_up) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; }   // a ray the filter cannot take: the frame goes to the last resort
                A.f->s_occluded[r] = W.have ? 1 : 0;
                state = BVH_FETCH;
            }
            uint32_t claimed = 0;
            if (CHUNKED) claimed = wave_fetch_chunked(cursor, lane, state == BVH_FETCH, chunk, A.chunk, total);
            if (state == BVH_FETCH) {   // (one fetch per free lane and round, as in bvh_shade_level)
                r = CHUNKED ? claimed : wave_fetch(cursor, lane);
                if (r >= total) state = BVH_OUT;
                else {
                    r += first;
                    const float4 q0 = A.f->s_shadowq[2 * (size_t)r], q1 = A.f->s_shadowq[2 * (size_t)r + 1];
                    // an unused slot (a level-0 pixel without a diffuse hit) needs nothing.  A light behind the surface contributes +-0
                    // times the albedo: no walk (kernel_plan.h has the argument)
                    if (__float_as_uint(q0.w) == SHADOW_SLOT_UNUSED) {}
                    else if (q1.w == 0.0f) A.f->s_occluded[r] = 0;
                    else {
                        R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                        R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;  // already normalised once; shadow rays skip shootRay (RayTracer.cpp:313-317)
                        ray_prepare(R);
                        light_dist = q0.w;
                        // an occluding hit has length(o + d t - o) <= light_dist with |d| = 1 up to two unit roundoffs: t <= light_dist (1 + 2^-16) (NaN: no bound)
                        bvh_walk_begin(W, light_dist * (1.0f + 0x1p-16f));
                        if (bvh_ray_setup(A, R, W.B)) state = BVH_WALK;
                        else { W.give_up = true; state = BVH_FINISHED; }
                    }
                }
            }
        }
        if (!__ballot(state != BVH_OUT)) break;
        if (state == BVH_WALK) {
            for (int it = 0; it < BVH_STEPS; ++it)
                if (state == BVH_WALK && !bvh_step<1, MODE>(A, R, false, light_dist, every_mesh, W, stack, nbox, ntri)) state = BVH_FINISHED;
        }
    }
    exec_counters_flush(A, nbox, ntri, lane);
}
template <uint32_t pass, int MODE>  // (the passes are kernels of their own in a profile)
__global__ __launch_bounds__(BLOCK) void bvh_trace_shadow(const KernelArgs A) {
    __shared__ uint32_t stack_lds[BVH_LDS_STACK * BLOCK];
    const uint32_t split = A.f->s_counts[SC_SHADOW_SPLIT];
    bvh_shadow_rays<MODE, pass == 0>(A, pass == 0 ? 0u : split, pass == 0 ? split : A.f->s_counts[SC_SHADOW] - split,
                          A.f->s_counts + (pass == 0 ? SC_SHADOW_FETCH : SC_SHADOW_FETCH2), stack_lds);
}

