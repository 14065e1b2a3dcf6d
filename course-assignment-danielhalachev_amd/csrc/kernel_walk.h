// kernel_walk.h -- the lean per-lane walk used by the ray-stream kernels when they are not counting.
//
// Same visits, same tests, same order as traversal_step (kernel_lane.h); what differs is the shape of the
// loop, which is what a wavefront of 64 unrelated rays pays for on every step:
//   * three mutually exclusive blocks per iteration -- one triangle, or one node, or (rarely) the end of a
//     mesh / the top-level tree -- instead of a five-way state machine;
//   * the triangle and node blocks are straight-line code: every early `return false` of the reference's
//     tests becomes a predicate, so there is nothing to diverge on inside a block;
//   * 32-bit byte offsets from a uniform base (one VGPR per address instead of 64-bit arithmetic);
//   * no parallel-axis handling in the box test: rays with a direction component below FLT_EPSILON
//     (BoundingBox.h:90) never enter this loop -- the caller sends them to heavy_trace.
#pragma once

#include "kernel_common.h"

// Early exit of a shadow walk, exact.  The reference walks every mesh to the end and then asks whether the
// mesh's closest hit (smallest t, KDTree.cpp:75-86) lies within the light's distance
// (AccelerationStructure.cpp:73-74: length(hitPoint - origin) <= distanceToLight), the answer being "any mesh
// says yes".  Every step of that length -- d*t, o + (.), (.) - o, the squares, their sums, the square root --
// is monotone non-decreasing in t >= 0 under round-to-nearest, so once ANY accepted hit with a finite t passes
// the comparison, the mesh's final closest hit (t no larger, and finite, because a finite distance always
// replaces a NaN/inf first entry) passes it too: the ray is occluded whatever the rest of the walk finds.
// hx, hy, hz = o + d*t as the triangle test computed it (Ray.cpp:21), which is the hitPoint the reference stores.
__device__ __forceinline__ bool shadow_hit_occludes(const Ray &R, float hx, float hy, float hz, float light_dist) {
    return len3(hx - R.ox, hy - R.oy, hz - R.oz) <= light_dist;
}

// The top-level tree for the per-lane walks, in LDS.  A ray spends about as many steps in the few dozen top-level nodes as in
// the mesh trees (HW14: 17 of 31 box tests of a shadow ray), and every one of them is a divergent gather through the vector
// L1 for a table of a kilobyte: when crt_create finds the table small (KernelArgs::top_fast) each workgroup copies it to LDS
// and the walks read nodes, leaf entries and mesh records from there (up to 256 nodes, 1024 leaf entries, 256 meshes).
typedef float lds_v4f __attribute__((ext_vector_type(4)));       // plain vector types: LDS pointers cannot bind to the float4 class
typedef uint32_t lds_v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lds_v4f *lds_f4;
typedef const __attribute__((address_space(3))) lds_v4u *lds_u4;
typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
constexpr int TOP_LDS_NODES = 256, TOP_LDS_ENTRIES = 1024, TOP_LDS_MESHES = 256;  // 16 KB per workgroup (KernelArgs::top_lds)
struct TopLdsStorage {
    lds_v4f nodes[TOP_LDS_NODES * 2];
    lds_v4u meshes[TOP_LDS_MESHES];    // {flags, tree root, quad root, -}
    uint32_t entries[TOP_LDS_ENTRIES];
};
struct TopLds {
    bool fast;        // the same for every thread of the launch
    lds_f4 nodes;     // node i at nodes[2 * (i - first)]
    lds_u4 meshes;
    lds_u32 entries;
    uint32_t first;
};
// Called by every thread of the workgroup before anything else (it holds a barrier).
__device__ __forceinline__ TopLds top_lds_load(const KernelArgs &A, TopLdsStorage &S) {
    TopLds T;
    T.fast = A.top_lds != 0;
    T.first = A.top_first;
    T.nodes = (lds_f4)S.nodes;
    T.meshes = (lds_u4)S.meshes;
    T.entries = (lds_u32)S.entries;
    if (T.fast) {
        for (uint32_t i = threadIdx.x; i < A.top_count * 2u; i += blockDim.x) {
            const float4 q = A.nodes[2 * (size_t)A.top_first + i];
            S.nodes[i] = lds_v4f{q.x, q.y, q.z, q.w};
        }
        for (uint32_t i = threadIdx.x; i < A.top_leaf_entries; i += blockDim.x) S.entries[i] = A.leaf_meshes[i];
        for (uint32_t i = threadIdx.x; i < A.top_meshes; i += blockDim.x) {
            const crt_mesh m = A.meshes[i];
            S.meshes[i] = lds_v4u{m.flags, m.root, A.quad_roots[i], m.pad};  // (pad: index + 1 among the single-leaf meshes, kernel_heavy.h)
        }
        __syncthreads();
    }
    return T;
}

struct LeanWalk {
    uint32_t n;        // next mesh-tree node (END: the current mesh is finished / none started)
    uint32_t e;        // next leaf entry of the current leaf (NONE: not inside a leaf)
    uint32_t tnode;    // next top-level node
    uint32_t tleaf;    // next top-level leaf entry (NONE: not inside a top-level leaf)
    uint32_t mesh;     // current mesh (NONE: none)
    float mmin, mt;    // mesh level: KDTree.cpp:75-86
    uint32_t mtri;
    bool mhave;
    float tmin, bt;    // scene level: KDTree.cpp:156-167
    uint32_t btri, bmesh;
    bool have, occluded;
    float light_dist;
    SeenMeshes seen;          // meshes 0..127 already walked for this ray (mesh_walk_is_repeat)
    uint32_t nbox, ntri;      // tests executed by this lane since the kernel started (KernelArgs::exec_count; not reset per ray)
};

__device__ __forceinline__ void lean_begin(LeanWalk &W, uint32_t top_root) {
    W.n = END; W.e = NONE; W.tnode = top_root; W.tleaf = NONE; W.mesh = NONE;
    W.mhave = false; W.mmin = INFINITY; W.mt = 0; W.mtri = 0;
    W.have = false; W.occluded = false; W.tmin = INFINITY; W.bt = 0; W.btri = 0; W.bmesh = 0;
    seen_clear(W.seen);
}

// Runs up to `iterations` steps of the walk for this lane.  Returns true when the ray's walk is complete.
template <bool SHADOW>
__device__ __forceinline__ bool lean_walk(LeanWalk &W, const Ray &R, const bool primary, const KernelArgs &A, const TopLds &TL,
                                          int iterations, uint32_t &steps) {
    const char *nodes_b = reinterpret_cast<const char *>(A.nodes);
    const char *ltris_b = reinterpret_cast<const char *>(A.ltris);
    for (; iterations > 0; --iterations) {
        steps++;
        if (W.e != NONE) {
            // ---- one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
            const float4 *T = reinterpret_cast<const float4 *>(ltris_b + (size_t)(uint32_t)(W.e << 6));
            const float4 a = T[0], b = T[1], c = T[2], d = T[3];
            if (A.exec_count) W.ntri++;
            const float nx = a.w, ny = b.w, nz = c.w;
            const float nd = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
            const float t = -(dot3(nx, ny, nz, R.ox, R.oy, R.oz) + d.x) / nd;
            const float px = R.ox + R.dx * t, py = R.oy + R.dy * t, pz = R.oz + R.dz * t;
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
            const bool ok = !(primary && nd >= 0) && !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
            // `closest = hits[0]; min = inf; for h: if (h.d < min) { min = h.d; closest = h; }`
            const bool less = ok && (t < W.mmin);
            const bool take = less || (ok && !W.mhave);
            W.mt = take ? t : W.mt;
            W.mtri = take ? __float_as_uint(d.y) : W.mtri;
            W.mmin = less ? t : W.mmin;
            W.mhave = W.mhave || ok;
            W.e = __float_as_uint(d.z) ? NONE : W.e + 1;
            if (SHADOW && ok && t < INFINITY && shadow_hit_occludes(R, px, py, pz, W.light_dist)) { W.occluded = true; return true; }
        } else if (W.n != END) {
            // ---- one mesh-tree node (KDTree.cpp:53-74, BoundingBox.h:85-108), branch-free
            const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(W.n << 5));
            const float4 q0 = N[0], q1 = N[1];
            if (A.exec_count) W.nbox++;
            const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
            const bool hit = slab_test_no_parallel(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
            const bool leaf = is_leaf_link(link);
            W.e = (hit && leaf) ? (link & ~LEAF) : NONE;
            W.n = (hit && !leaf) ? link : miss;
        } else {
            // ---- rare: a mesh ended, or the walk is in the top-level tree (KDTree.cpp:132-167, AccelerationStructure.cpp:60-90)
            if (W.mesh != NONE) {
                if (W.mhave) {
                    if (SHADOW) {
                        const float px = R.ox + R.dx * W.mt, py = R.oy + R.dy * W.mt, pz = R.oz + R.dz * W.mt;
                        if (len3(px - R.ox, py - R.oy, pz - R.oz) <= W.light_dist) W.occluded = true;
                    } else {
                        if (!W.have) { W.have = true; W.bt = W.mt; W.btri = W.mtri; W.bmesh = W.mesh; }
                        if (W.mt < W.tmin) { W.tmin = W.mt; W.bt = W.mt; W.btri = W.mtri; W.bmesh = W.mesh; }
                    }
                }
                W.mesh = NONE;
            }
            if (W.tleaf != NONE) {
                const uint32_t ent = TL.fast ? TL.entries[W.tleaf] : A.leaf_meshes[W.tleaf];
                const uint32_t mi = ent & ~LAST;
                W.tleaf = (ent & LAST) ? NONE : W.tleaf + 1;
                uint4 m;  // {flags, tree root, quad root}
                if (TL.fast) { const lds_v4u v = TL.meshes[mi]; m = make_uint4(v[0], v[1], v[2], v[3]); }
                else { const crt_mesh g = A.meshes[mi]; m = make_uint4(g.flags, g.root, 0u, 0u); }
                if (!(SHADOW && (m.x & 1u)) && !mesh_walk_is_repeat(W.seen, mi)) {
                    W.mesh = mi;
                    W.n = m.y;
                    W.mhave = false;
                    W.mmin = INFINITY;
                }
            } else if (W.tnode != END) {
                float4 q0, q1;
                if (TL.fast) {
                    const lds_v4f a = TL.nodes[2 * (W.tnode - TL.first)], b = TL.nodes[2 * (W.tnode - TL.first) + 1];
                    q0 = make_float4(a[0], a[1], a[2], a[3]); q1 = make_float4(b[0], b[1], b[2], b[3]);
                }
                else {
                    const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(W.tnode << 5));
                    q0 = N[0]; q1 = N[1];
                }
                if (A.exec_count) W.nbox++;
                const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
                const bool hit = slab_test_no_parallel(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
                const bool leaf = is_leaf_link(link);
                W.tleaf = (hit && leaf) ? (link & ~LEAF) : NONE;
                W.tnode = (hit && !leaf) ? link : miss;
            } else {
                return true;
            }
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// The same walk over QUAD nodes.
//
// crt_create collapses every mesh tree into nodes of up to four slots (crt_device.hip: build_quads): a slot
// is a descendant of the binary node the quad stands for -- its children, with the largest inner ones replaced
// by THEIR children until four slots are filled -- listed in the reference's visit order (children[1] before
// children[0], KDTree.cpp:64-72).  A slot is entered iff its own box passes; because every child box lies
// inside its parent's box and the slab test is monotone under that inclusion (DESIGN.md "nesting"), that is
// exactly when the reference's walk reaches it, and a depth-first walk over slots in order meets the leaves in
// the reference's order -- so hits are collected in the same order and "first collected wins ties" holds.
// What changes is the cost: one 128-byte record (one cache line) and one loop trip test four boxes, where the
// binary layout spends four dependent 32-byte gathers; the pending slots wait on a per-lane stack in LDS.
// The walk needs nested boxes; crt_create only selects it when it has verified them.
// The stack holds A.quad_stack_depth words per lane (LDS: depth x BLOCK x 4 bytes per workgroup); a walk that
// would need more is restarted in the wave-per-ray kernel, which keeps no stack.

struct QuadWalk {
    uint32_t q;        // quad to test next (NONE: none)
    uint32_t e;        // next leaf entry of the current leaf (NONE: not inside a leaf)
    uint32_t sp;       // pending slots: `top` + sp-1 words in LDS
    uint32_t top;
    uint32_t tnode, tleaf, mesh;
    float mmin, mt;
    uint32_t mtri;
    bool mhave;
    float tmin, bt;
    uint32_t btri, bmesh;
    bool have, occluded;
    float light_dist;
    Prune prune;       // closest-hit walks (kernel_common.h: distance pruning)
    SeenMeshes seen;          // meshes 0..127 already walked for this ray (mesh_walk_is_repeat)
    uint32_t nbox, ntri;      // tests executed by this lane since the kernel started (KernelArgs::exec_count; not reset per ray)
};

__device__ __forceinline__ void lean_begin(QuadWalk &W, uint32_t top_root) {
    W.q = NONE; W.e = NONE; W.sp = 0; W.top = NONE; W.tnode = top_root; W.tleaf = NONE; W.mesh = NONE;
    W.mhave = false; W.mmin = INFINITY; W.mt = 0; W.mtri = 0;
    W.have = false; W.occluded = false; W.tmin = INFINITY; W.bt = 0; W.btri = 0; W.bmesh = 0;
    seen_clear(W.seen);
}

enum : int { WALK_MORE = 0, WALK_DONE = 1, WALK_STACK_FULL = 2 };

// `stk`: this thread's column of the workgroup's stack array (element i at stk[i * BLOCK]).
template <bool SHADOW>
__device__ __forceinline__ int quad_walk(QuadWalk &W, const Ray &R, const bool primary, const KernelArgs &A, const TopLds &TL,
                                         int iterations, uint32_t &steps, uint32_t *stk) {
    const char *nodes_b = reinterpret_cast<const char *>(A.nodes);
    const char *quads_b = reinterpret_cast<const char *>(A.quads);
    const char *ltris_b = reinterpret_cast<const char *>(A.ltris);
    for (; iterations > 0; --iterations) {
        steps++;
        bool next = false;
        uint32_t cur = NONE;
        if (W.e != NONE) {
            // ---- one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
            const float4 *T = reinterpret_cast<const float4 *>(ltris_b + (size_t)(uint32_t)(W.e << 6));
            const float4 a = T[0], b = T[1], c = T[2], d = T[3];
            if (A.exec_count) W.ntri++;
            const float nx = a.w, ny = b.w, nz = c.w;
            const float nd = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
            const float t = -(dot3(nx, ny, nz, R.ox, R.oy, R.oz) + d.x) / nd;
            const float px = R.ox + R.dx * t, py = R.oy + R.dy * t, pz = R.oz + R.dz * t;
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
            const bool ok = !(primary && nd >= 0) && !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
            const bool less = ok && (t < W.mmin);
            const bool take = less || (ok && !W.mhave);
            W.mt = take ? t : W.mt;
            W.mtri = take ? __float_as_uint(d.y) : W.mtri;
            W.mmin = less ? t : W.mmin;
            W.mhave = W.mhave || ok;
            next = __float_as_uint(d.z) != 0;
            W.e = next ? NONE : W.e + 1;
            if (SHADOW && ok && t < INFINITY && shadow_hit_occludes(R, px, py, pz, W.light_dist)) { W.occluded = true; return WALK_DONE; }
        } else if (W.q != NONE) {
            // ---- one quad: four boxes (BoundingBox.h:85-108), the slots that pass go on the stack, last first
            if (W.sp + 4 > A.quad_stack_depth) return WALK_STACK_FULL;
            const float4 *Q = reinterpret_cast<const float4 *>(quads_b + (size_t)(uint32_t)(W.q << 8));
            const float4 lx = Q[0], ly = Q[1], lz = Q[2], hx = Q[3], hy = Q[4], hz = Q[5];
            const uint4 lk = reinterpret_cast<const uint4 *>(Q)[6];
            if (A.exec_count) W.nbox += (lk.x != NONE) + (lk.y != NONE) + (lk.z != NONE) + (lk.w != NONE);
            bool h0 = lk.x != NONE && slab_test_no_parallel(R, lx.x, ly.x, lz.x, hx.x, hy.x, hz.x);
            bool h1 = lk.y != NONE && slab_test_no_parallel(R, lx.y, ly.y, lz.y, hx.y, hy.y, hz.y);
            bool h2 = lk.z != NONE && slab_test_no_parallel(R, lx.z, ly.z, lz.z, hx.z, hy.z, hz.z);
            bool h3 = lk.w != NONE && slab_test_no_parallel(R, lx.w, ly.w, lz.w, hx.w, hy.w, hz.w);
            if (!SHADOW && (A.prune & 1u)) {
                const float bound = fminf(W.tmin, W.mmin);  // only a strictly smaller distance changes anything
                if (bound < INFINITY) {
                    const float4 cx = Q[(W.prune.bits & 8u) ? 8 : 11], cy = Q[(W.prune.bits & 16u) ? 9 : 12], cz = Q[(W.prune.bits & 32u) ? 10 : 13];
                    h0 = h0 && !(prune_bound(W.prune, R, cx.x, cy.x, cz.x) >= bound);
                    h1 = h1 && !(prune_bound(W.prune, R, cx.y, cy.y, cz.y) >= bound);
                    h2 = h2 && !(prune_bound(W.prune, R, cx.z, cy.z, cz.z) >= bound);
                    h3 = h3 && !(prune_bound(W.prune, R, cx.w, cy.w, cz.w) >= bound);
                }
            }
            // push(x): the old top goes to LDS, x becomes the top
#define CRT_QPUSH(x)                                                   \
    do {                                                               \
        if (W.sp > 0) stk[(W.sp - 1) * BLOCK] = W.top;                 \
        W.top = (x);                                                   \
        W.sp++;                                                        \
    } while (0)
            if (h3) cur = lk.w;
            if (h2) { if (cur != NONE) CRT_QPUSH(cur); cur = lk.z; }
            if (h1) { if (cur != NONE) CRT_QPUSH(cur); cur = lk.y; }
            if (h0) { if (cur != NONE) CRT_QPUSH(cur); cur = lk.x; }
#undef CRT_QPUSH
            W.q = NONE;
            next = true;
        } else {
            // ---- rare: a mesh ended, or the walk is in the top-level tree (KDTree.cpp:132-167, AccelerationStructure.cpp:60-90)
            if (W.mesh != NONE) {
                if (W.mhave) {
                    if (SHADOW) {
                        const float px = R.ox + R.dx * W.mt, py = R.oy + R.dy * W.mt, pz = R.oz + R.dz * W.mt;
                        if (len3(px - R.ox, py - R.oy, pz - R.oz) <= W.light_dist) W.occluded = true;
                    } else {
                        if (!W.have) { W.have = true; W.bt = W.mt; W.btri = W.mtri; W.bmesh = W.mesh; }
                        if (W.mt < W.tmin) { W.tmin = W.mt; W.bt = W.mt; W.btri = W.mtri; W.bmesh = W.mesh; }
                    }
                }
                W.mesh = NONE;
            }
            if (W.tleaf != NONE) {
                const uint32_t ent = TL.fast ? TL.entries[W.tleaf] : A.leaf_meshes[W.tleaf];
                const uint32_t mi = ent & ~LAST;
                W.tleaf = (ent & LAST) ? NONE : W.tleaf + 1;
                uint4 m;  // {flags, tree root, quad root}
                if (TL.fast) { const lds_v4u v = TL.meshes[mi]; m = make_uint4(v[0], v[1], v[2], v[3]); }
                else { m = make_uint4(A.meshes[mi].flags, 0u, A.quad_roots[mi], 0u); }
                if (!(SHADOW && (m.x & 1u)) && !mesh_walk_is_repeat(W.seen, mi)) {
                    W.mesh = mi;
                    W.q = m.z;
                    W.mhave = false;
                    W.mmin = INFINITY;
                }
            } else if (W.tnode != END) {
                float4 q0, q1;
                if (TL.fast) {
                    const lds_v4f a = TL.nodes[2 * (W.tnode - TL.first)], b = TL.nodes[2 * (W.tnode - TL.first) + 1];
                    q0 = make_float4(a[0], a[1], a[2], a[3]); q1 = make_float4(b[0], b[1], b[2], b[3]);
                }
                else {
                    const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(W.tnode << 5));
                    q0 = N[0]; q1 = N[1];
                }
                if (A.exec_count) W.nbox++;
                const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
                const bool hit = slab_test_no_parallel(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
                const bool leaf = is_leaf_link(link);
                W.tleaf = (hit && leaf) ? (link & ~LEAF) : NONE;
                W.tnode = (hit && !leaf) ? link : miss;
            } else {
                return WALK_DONE;
            }
        }
        if (next) {
            // the next slot: the first one that passed just now, else the most recent pending one
            if (cur == NONE && W.sp > 0) {
                cur = W.top;
                W.sp--;
                if (W.sp > 0) W.top = stk[(W.sp - 1) * BLOCK];
            }
            if (cur != NONE) {
                if (cur & LEAF) W.e = cur & ~LEAF;
                else W.q = cur;
            }
        }
    }
    return WALK_MORE;
}

// the binary walk under the same interface
template <bool SHADOW>
__device__ __forceinline__ int quad_walk(LeanWalk &W, const Ray &R, const bool primary, const KernelArgs &A, const TopLds &TL,
                                         int iterations, uint32_t &steps, uint32_t *) {
    return lean_walk<SHADOW>(W, R, primary, A, TL, iterations, steps) ? WALK_DONE : WALK_MORE;
}

// End of a kernel: the lanes' executed-test tallies go to KernelArgs::exec_counters (collect_counters == 2 only).
// `nplan`: box tests of the plan loops (kernel_plan.h) -- executed per lane, but their operands are scalar registers loaded once
// per wave, so they are tallied apart from the tests that fetch a node per lane.
__device__ __forceinline__ void exec_counters_flush(const KernelArgs &A, uint32_t nbox, uint32_t ntri, uint32_t lane, uint32_t nplan = 0) {
    if (!A.exec_count) return;
    unsigned long long b = nbox, t = ntri, p = nplan;
    for (int off = 32; off > 0; off >>= 1) { b += __shfl_down(b, off); t += __shfl_down(t, off); p += __shfl_down(p, off); }
    if (lane == 0) { atomicAdd(&A.exec_counters[0], b); atomicAdd(&A.exec_counters[1], t); if (p) atomicAdd(A.exec_plan, p); }
}
