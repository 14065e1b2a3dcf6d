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

// End of a kernel: the lanes' executed-test tallies go to KernelArgs::exec_counters (collect_counters == 2 only).
// `nplan`: box tests of the plan loops (kernel_plan.h) -- executed per lane, but their operands are scalar registers loaded once
// per wave, so they are tallied apart from the tests that fetch a node per lane.
__device__ __forceinline__ void exec_counters_flush(const KernelArgs &A, uint32_t nbox, uint32_t ntri, uint32_t lane, uint32_t nplan = 0) {
    if (!A.exec_count) return;
    unsigned long long b = nbox, t = ntri, p = nplan;
    for (int off = 32; off > 0; off >>= 1) { b += __shfl_down(b, off); t += __shfl_down(t, off); p += __shfl_down(p, off); }
    if (lane == 0) { atomicAdd(&A.exec_counters[0], b); atomicAdd(&A.exec_counters[1], t); if (p) atomicAdd(A.exec_plan, p); }
}
