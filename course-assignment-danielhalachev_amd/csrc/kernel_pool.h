// kernel_pool.h -- the planned shadow walk (kernel_plan.h) with a POOL of rays per lane, so that a loop trip runs ONE block
// with most lanes taking part.
//
// In stream_trace_shadow_plan a lane holds one ray; at any moment about a third of the lanes of a wave want a triangle test
// and two thirds a node test, a trip runs both blocks, and on average 24 of 64 lanes take part in an instruction (PMC,
// DESIGN.md section 4).  Letting lanes wait until enough of them want the same block only makes the others wait
// (tri_gather: measured slower).  Here every lane owns POOL_RAYS = 3 rays instead: one in registers, the others parked in
// LDS at lane-private addresses (no cross-lane traffic, no synchronisation between waves).  The wave is in NODE mode or in
// TRIANGLE mode; in each trip every lane brings one of ITS rays that wants the mode's block into registers (four 16-byte LDS
// writes and reads when it has to swap) and the wave runs that one block.  With three rays to choose from, a lane has a
// node-wanting ray 96 % of the time and a triangle-wanting one 72 % of the time; the wave changes mode when too few lanes
// can take part.  Every ray is walked exactly as kernel_plan.h walks it -- the same node steps and triangle tests in the
// same order -- so the flags are the same bit for bit; only the interleaving of different rays changes.
#pragma once

#include "kernel_plan.h"

constexpr int POOL_RAYS = 3;                 // rays per lane: one in registers, POOL_RAYS - 1 parked
constexpr int POOL_PARKED = POOL_RAYS - 1;
constexpr uint32_t POOL_LDS_BYTES = 4u * POOL_PARKED * BLOCK * 16u;  // four 16-byte groups per parked ray
enum : uint32_t { PM_EMPTY = 0, PM_NODE = 1, PM_TRI = 2 };

struct PoolRay {           // a shadow ray in flight, 16 dwords
    Ray R;                 // (parmask is always 0 here: other rays go to the wave-per-ray kernel)
    float dist;            // distance to the light
    uint32_t wn, we;       // next mesh-tree node, leaf cursor (kernel_plan.h)
    uint32_t mlo, mhi;     // meshes still to walk
    uint32_t r, steps;     // shadow-queue slot, steps taken so far
};

typedef float pool_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pool_store(pool_v4 *lds, const uint32_t k, const PoolRay &P) {
    const uint32_t t = threadIdx.x;
    lds[(0 * POOL_PARKED + k) * BLOCK + t] = pool_v4{P.R.ox, P.R.oy, P.R.oz, P.dist};
    lds[(1 * POOL_PARKED + k) * BLOCK + t] = pool_v4{P.R.dx, P.R.dy, P.R.dz, __uint_as_float(P.we)};
    lds[(2 * POOL_PARKED + k) * BLOCK + t] = pool_v4{P.R.ix, P.R.iy, P.R.iz, __uint_as_float(P.wn)};
    lds[(3 * POOL_PARKED + k) * BLOCK + t] = pool_v4{__uint_as_float(P.mlo), __uint_as_float(P.mhi), __uint_as_float(P.r), __uint_as_float(P.steps)};
}
__device__ __forceinline__ void pool_load(const pool_v4 *lds, const uint32_t k, PoolRay &P) {
    const uint32_t t = threadIdx.x;
    const pool_v4 a = lds[(0 * POOL_PARKED + k) * BLOCK + t], b = lds[(1 * POOL_PARKED + k) * BLOCK + t];
    const pool_v4 c = lds[(2 * POOL_PARKED + k) * BLOCK + t], d = lds[(3 * POOL_PARKED + k) * BLOCK + t];
    P.R.ox = a[0]; P.R.oy = a[1]; P.R.oz = a[2]; P.dist = a[3];
    P.R.dx = b[0]; P.R.dy = b[1]; P.R.dz = b[2]; P.we = __float_as_uint(b[3]);
    P.R.ix = c[0]; P.R.iy = c[1]; P.R.iz = c[2]; P.wn = __float_as_uint(c[3]);
    P.mlo = __float_as_uint(d[0]); P.mhi = __float_as_uint(d[1]); P.r = __float_as_uint(d[2]); P.steps = __float_as_uint(d[3]);
    P.R.parmask = 0;
}

// the shadow rays [first, first + total) of the queue; `cursor` hands them out
__device__ __forceinline__ void shadow_pool_walks(const KernelArgs &A, const uint32_t first, const uint32_t total, uint32_t *cursor) {
    extern __shared__ pool_v4 pool_lds[];         // parked rays
    __shared__ uint32_t root_of_bit[64];          // shadow order -> root node of the mesh's tree
    if (threadIdx.x < 64u) root_of_bit[threadIdx.x] = threadIdx.x < A.plan_shadow_bits ? A.meshes[A.plan_shadow_mesh[threadIdx.x]].root : END;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    if (A.s_counts[SC_OVERFLOW]) return;
    const char *nodes_b = reinterpret_cast<const char *>(A.pnodes);
    const char *ptris_b = reinterpret_cast<const char *>(A.ptris);

    PoolRay P;                          // the ray in registers
    P.R.ox = P.R.oy = P.R.oz = P.R.dx = P.R.dy = P.R.dz = P.R.ix = P.R.iy = P.R.iz = 0; P.R.parmask = 0;
    P.dist = 0; P.wn = END; P.we = NONE; P.mlo = P.mhi = 0; P.r = 0; P.steps = 0;
    uint32_t am = PM_EMPTY;             // what the ray in registers wants next
    uint32_t pm[POOL_PARKED];           // ... and the parked ones
    for (int k = 0; k < POOL_PARKED; k++) pm[k] = PM_EMPTY;
    uint32_t nbox = 0, ntri = 0, nplan = 0;
    bool exhausted = false;             // the queue has no more rays (wave-uniform)
    uint32_t mode = PM_NODE;            // the wave's mode (wave-uniform)
    bool starving = true;               // few lanes could take part in the last trip: look at the queue before the next one
    const uint32_t guard_trips = 1u << 24;  // every ray ends within its step budget; this bound is never reached

    for (uint32_t trip = 0; trip < guard_trips; trip++) {
        // ---- 1. refill: a round gives every lane that has an empty place one new ray (plan: a wave-uniform loop, so rounds
        //         are only worth it for many lanes at once -- or when the wave would otherwise run dry)
        if (!exhausted && ((trip & 7u) == 0u || starving)) {
            for (int round = 0; round < POOL_RAYS; round++) {
                bool has_empty = am == PM_EMPTY;
                for (int k = 0; k < POOL_PARKED; k++) has_empty = has_empty || pm[k] == PM_EMPTY;
                const uint32_t wanting = (uint32_t)__popcll(__ballot(has_empty));
                bool any_ray = am != PM_EMPTY;
                for (int k = 0; k < POOL_PARKED; k++) any_ray = any_ray || pm[k] != PM_EMPTY;
                const bool dry = __ballot(any_ray) == 0ull;
                if (!(wanting >= A.pool_refill || (dry && wanting))) break;
                bool fresh = false;
                PoolRay N;
                N.R.parmask = 0; N.dist = 0; N.wn = END; N.we = NONE; N.mlo = N.mhi = 0; N.r = 0; N.steps = 0;
                N.R.ox = N.R.oy = N.R.oz = N.R.dx = N.R.dy = N.R.dz = N.R.ix = N.R.iy = N.R.iz = 0;
                bool asking = has_empty;
                while (asking) {
                    uint32_t r = wave_fetch(cursor, lane);
                    if (r >= total) { exhausted = true; break; }
                    r += first;
                    const float4 q0 = A.s_shadowq[2 * (size_t)r], q1 = A.s_shadowq[2 * (size_t)r + 1];
                    if (__float_as_uint(q0.w) == SHADOW_SLOT_UNUSED) continue;  // a level-0 pixel without a diffuse hit
                    N.R.ox = q0.x; N.R.oy = q0.y; N.R.oz = q0.z;
                    N.R.dx = q1.x; N.R.dy = q1.y; N.R.dz = q1.z;  // already normalised once; shadow rays skip shootRay (RayTracer.cpp:313-317)
                    ray_prepare(N.R);
                    if (N.R.parmask != 0) {  // BoundingBox.h:90-93 needs the general test: the wave-per-ray kernel has it
                        if (evict_ray(A.s_sheavy, A.s_heavy_cap, A.s_counts + SC_SHEAVY, r, lane)) continue;
                        A.s_counts[SC_OVERFLOW] = 1;
                        continue;
                    }
                    N.dist = q0.w;
                    N.r = r;
                    fresh = true;
                    asking = false;
                }
                exhausted = __ballot(exhausted) != 0ull;
                if (fresh) {
                    // the plan: which meshes does this ray have to walk?  (k is wave-uniform: scalar loads, no gathers)
                    uint32_t lo = 0, hi = 0;
                    for (uint32_t k = 0; k < A.plan_leaves; k++) {
                        const v16f L = plan_leaf(A, k);
                        const bool hit = slab_test_no_parallel(N.R, L[0], L[1], L[2], L[4], L[5], L[6]);
                        lo |= hit ? __float_as_uint(L[8]) : 0u;
                        hi |= hit ? __float_as_uint(L[9]) : 0u;
                    }
                    if (A.exec_count) nplan += A.plan_leaves;
                    N.mlo = lo; N.mhi = hi;
                    // into the registers if they are free, else into the first free parked place
                    if (am == PM_EMPTY) { P = N; am = PM_NODE; }
                    else {
                        bool placed = false;
                        for (int k = 0; k < POOL_PARKED; k++)
                            if (!placed && pm[k] == PM_EMPTY) { pool_store(pool_lds, (uint32_t)k, N); pm[k] = PM_NODE; placed = true; }
                    }
                }
                if (exhausted) break;
            }
        }
        // ---- 2. the wave's mode: stay while enough lanes can take part, else go where more can
        bool can_node = am == PM_NODE, can_tri = am == PM_TRI;
        for (int k = 0; k < POOL_PARKED; k++) { can_node = can_node || pm[k] == PM_NODE; can_tri = can_tri || pm[k] == PM_TRI; }
        const uint32_t n_node = (uint32_t)__popcll(__ballot(can_node)), n_tri = (uint32_t)__popcll(__ballot(can_tri));
        if (n_node == 0u && n_tri == 0u && exhausted) break;  // nothing in flight, nothing to fetch: done
        // (nothing in flight but rays left in the queue: `starving` sends the next trip to the refill; the rest of this one idles)
        const uint32_t n_mine = mode == PM_NODE ? n_node : n_tri, n_other = mode == PM_NODE ? n_tri : n_node;
        if (n_mine == 0u || (n_mine < A.pool_switch && n_other > n_mine)) mode = mode == PM_NODE ? PM_TRI : PM_NODE;
        starving = (n_node + n_tri) < 64u;
        // ---- 3. every lane brings one of its rays that wants this block into the registers
        const bool swap = am != mode && (mode == PM_NODE ? can_node : can_tri);
        if (__ballot(swap)) {
            if (swap) {
                int pick = -1;
                for (int k = POOL_PARKED - 1; k >= 0; k--) if (pm[k] == mode) pick = k;
                PoolRay Q;
                pool_load(pool_lds, (uint32_t)pick, Q);
                if (am != PM_EMPTY) pool_store(pool_lds, (uint32_t)pick, P);
                const uint32_t was = am;
                P = Q;
                am = mode;
                for (int k = 0; k < POOL_PARKED; k++) if (k == pick) pm[k] = was;
            }
        }
        // ---- 4. the block
        if (am == mode) {
            P.steps++;
            bool finished = false, occluded = false;
            if (mode == PM_TRI) {
                // one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
                const float4 *T = reinterpret_cast<const float4 *>(ptris_b + (size_t)(leaf_cursor_entry(P.we) * 48u));
                const float4 a = T[0], b = T[1], c = T[2];
                if (A.exec_count) ntri++;
                const Ray &R = P.R;
                const float nx = a.w, ny = b.w, nz = c.w;
                const float nd = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
                const float plane = -dot3(a.x, a.y, a.z, nx, ny, nz);  // distanceToPlane, Ray.cpp:17
                const float t = -(dot3(nx, ny, nz, R.ox, R.oy, R.oz) + plane) / nd;
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
                // shadow rays are not culled (Ray.cpp:13 is PrimaryRay only)
                const bool ok = !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
                P.we = leaf_cursor_next(P.we);
                if (ok && t < INFINITY && shadow_hit_occludes(R, px, py, pz, P.dist)) { finished = true; occluded = true; }
                else if (P.we == NONE) am = PM_NODE;  // the leaf is done: back to the nodes
            } else {
                // one mesh-tree node (KDTree.cpp:53-74, BoundingBox.h:85-108), branch-free, the step to the next mesh folded in
                bool walk = true;
                if (P.wn == END) {
                    if ((P.mlo | P.mhi) == 0u) { finished = true; walk = false; }  // no mesh is left: not occluded
                    else {
                        const uint32_t bit = P.mlo ? (uint32_t)__builtin_ctz(P.mlo) : 32u + (uint32_t)__builtin_ctz(P.mhi);
                        if (P.mlo) P.mlo &= P.mlo - 1u; else P.mhi &= P.mhi - 1u;
                        P.wn = root_of_bit[bit];
                    }
                }
                if (walk) {
                    const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(P.wn << 5));
                    const float4 q0 = N[0], q1 = N[1];
                    if (A.exec_count) nbox++;
                    const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
                    const bool hit = slab_test_no_parallel(P.R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
                    const bool leaf = is_leaf_link(link);
                    P.we = (hit && leaf) ? (link & ~LEAF) : NONE;
                    P.wn = (hit && !leaf) ? link : miss;
                    if (P.we != NONE) am = PM_TRI;  // a leaf: its triangles next
                }
            }
            if (finished) {
                A.s_occluded[P.r] = occluded ? 1 : 0;
                am = PM_EMPTY;
            } else if (P.steps >= A.step_budget) {
                // a long walk goes to the wave-per-ray kernel, which restarts it (the list full: keep walking here)
                if (evict_ray(A.s_sheavy, A.s_heavy_cap, A.s_counts + SC_SHEAVY, P.r, lane)) am = PM_EMPTY;
                P.steps = 0;
            }
        }
    }
    exec_counters_flush(A, nbox, ntri, lane, nplan);
}

template <uint32_t pass>
__global__ __launch_bounds__(BLOCK) void stream_trace_shadow_pool(const KernelArgs A) {
    const uint32_t split = A.s_counts[SC_SHADOW_SPLIT];
    shadow_pool_walks(A, pass == 0 ? 0u : split, pass == 0 ? split : A.s_counts[SC_SHADOW] - split,
                      A.s_counts + (pass == 0 ? SC_SHADOW_FETCH : SC_SHADOW_FETCH2));
}

// one recursion level's shadow rays (kernel_plan.h: stream_trace_shadow_level)
__global__ __launch_bounds__(BLOCK) void stream_trace_shadow_pool_level(const KernelArgs A, const uint32_t gen) {
    const uint32_t begin = A.s_counts[SC_LSPLIT + gen], end = A.s_counts[SC_LSPLIT + gen + 1];
    shadow_pool_walks(A, begin, end > begin ? end - begin : 0u, A.s_counts + SC_LFETCH + gen);
}
