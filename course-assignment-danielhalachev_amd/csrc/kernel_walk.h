// kernel_walk.h -- what the production walk kernels (kernel_plan.h, kernel_heavy.h) share.
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
    lds_v4u meshes[TOP_LDS_MESHES];    // {flags, tree root, -, index + 1 among the single-leaf meshes}
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
    T.fast = A.s->top_lds != 0;
    T.first = A.s->top_first;
    T.nodes = (lds_f4)S.nodes;
    T.meshes = (lds_u4)S.meshes;
    T.entries = (lds_u32)S.entries;
    if (T.fast) {
        for (uint32_t i = threadIdx.x; i < A.s->top_count * 2u; i += blockDim.x) {
            const float4 q = A.s->nodes[2 * (size_t)A.s->top_first + i];
            S.nodes[i] = lds_v4f{q.x, q.y, q.z, q.w};
        }
        for (uint32_t i = threadIdx.x; i < A.s->top_leaf_entries; i += blockDim.x) S.entries[i] = A.s->leaf_meshes[i];
        for (uint32_t i = threadIdx.x; i < A.s->top_meshes; i += blockDim.x) {
            const crt_mesh m = A.s->meshes[i];
            S.meshes[i] = lds_v4u{m.flags, m.root, 0u, m.pad};  // (pad: index + 1 among the single-leaf meshes, kernel_heavy.h)
        }
        __syncthreads();
    }
    return T;
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
