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
};

__device__ __forceinline__ void lean_begin(LeanWalk &W, uint32_t top_root) {
    W.n = END; W.e = NONE; W.tnode = top_root; W.tleaf = NONE; W.mesh = NONE;
    W.mhave = false; W.mmin = INFINITY; W.mt = 0; W.mtri = 0;
    W.have = false; W.occluded = false; W.tmin = INFINITY; W.bt = 0; W.btri = 0; W.bmesh = 0;
}

// Runs up to `iterations` steps of the walk for this lane.  Returns true when the ray's walk is complete.
template <bool SHADOW>
__device__ __forceinline__ bool lean_walk(LeanWalk &W, const Ray &R, const bool primary, const KernelArgs &A,
                                          int iterations, uint32_t &steps) {
    const char *nodes_b = reinterpret_cast<const char *>(A.nodes);
    const char *ltris_b = reinterpret_cast<const char *>(A.ltris);
    for (; iterations > 0; --iterations) {
        steps++;
        if (W.e != NONE) {
            // ---- one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
            const float4 *T = reinterpret_cast<const float4 *>(ltris_b + (size_t)(uint32_t)(W.e << 6));
            const float4 a = T[0], b = T[1], c = T[2], d = T[3];
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
        } else if (W.n != END) {
            // ---- one mesh-tree node (KDTree.cpp:53-74, BoundingBox.h:85-108), branch-free
            const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(W.n << 5));
            const float4 q0 = N[0], q1 = N[1];
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
                const uint32_t ent = A.leaf_meshes[W.tleaf];
                const uint32_t mi = ent & ~LAST;
                W.tleaf = (ent & LAST) ? NONE : W.tleaf + 1;
                const crt_mesh m = A.meshes[mi];
                if (!(SHADOW && (m.flags & 1u))) {
                    W.mesh = mi;
                    W.n = m.root;
                    W.mhave = false;
                    W.mmin = INFINITY;
                }
            } else if (W.tnode != END) {
                const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(W.tnode << 5));
                const float4 q0 = N[0], q1 = N[1];
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
