// kernel_plan.h -- the per-lane ray-stream kernels with the top-level tree evaluated as a PLAN.
//
// What kernel_walk.h's walks spend most of their loop trips on is not the mesh trees but the top-level tree: on the
// benchmark scene (9 meshes, 39 top-level nodes, 20 leaves listing 75 entries) a ray takes ~17 node steps and ~37 entry
// steps up there -- each one a trip through the "rare" block of a divergent wave -- against ~14 node and ~9 triangle steps
// inside the meshes.  None of that depends on anything but the ray and a table of a kilobyte:
//
//  * the top-level tree's boxes are nested (crt_create checks it) and the slab test is monotone under that nesting, so the
//    reference reaches a top-level leaf exactly when the leaf's OWN box passes its exact test (the argument of
//    kernel_heavy.h for the mesh trees);
//  * so when a wave refills, the new rays are tested against leaf k for k = 0 .. leaves-1 in a WAVE-UNIFORM loop: the box
//    sits in scalar registers (one s_load per leaf for the whole wave), there is no gather and no divergence;
//  * a SHADOW ray only needs the SET of non-refractive meshes listed in the leaves it reaches
//    (AccelerationStructure.cpp:60-90 is an OR over them): a 64-bit mask, walked big meshes first (likeliest occluders);
//  * a CLOSEST-HIT ray needs them in the reference's order -- leaves in visit order, entries in list order, every mesh at
//    its first occurrence (kernel_common.h: mesh_walk_is_repeat) -- because ties between meshes go to the first one
//    collected (KDTree.cpp:159-166): the same uniform loop appends them to a per-lane byte list in LDS.
//
// What remains per lane is the walk of the mesh trees themselves: two blocks (one triangle / one node), the step from one
// mesh to the next folded into the node block.
//
// A shadow walk needs no per-mesh bookkeeping at all: the early exit (kernel_walk.h: shadow_hit_occludes) fires on the
// first accepted hit with a finite t within the light's distance, and the reference's end-of-mesh test on the mesh's
// closest hit can only succeed if that very test had succeeded when the hit was accepted (same floats, same expression;
// an infinite or NaN t fails both).  So "occluded" == "some accepted hit passes the distance test", mesh by mesh or not.
#pragma once

#include "kernel_common.h"
#include "kernel_stream.h"
#include "kernel_walk.h"

// one top-level leaf: {lo.xyz, first entry} {hi.xyz, entry count} {shadow mask words 0..3} {words 4..7}: 16 dwords, one s_load_dwordx16
// The WIDE plan (more than 64 top-level leaves or meshes, up to 256 meshes): the leaves in groups of PLAN_GROUP_LEAVES consecutive
// ones (visit order is spatial order) under their union box -- {lo.xyz, first leaf} {hi.xyz, leaves}, one s_load_dwordx8.  A ray
// that passes a leaf's box passes any box containing it (the slab test is monotone, kernel_heavy.h), so a group none of the wave's
// rays passes holds no leaf any of them passes: skipping it changes nothing.
typedef float v8f __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) v8f *kv8p;
__device__ __forceinline__ v8f plan_group(const KernelArgs &A, uint32_t g) {
    return *(kv8p)((kfp)(const float *)A.s->plan_groups + PLAN_GROUP_DWORDS * (size_t)g);
}

__device__ __forceinline__ v16f plan_leaf(const KernelArgs &A, uint32_t k) {
    return *(kv16p)((kfp)(const float *)A.s->plan_boxes + PLAN_LEAF_DWORDS * (size_t)k);
}
// the same leaf with the shadow masks of the table `boxes` (plan_boxes: the non-refractive meshes; plan_boxes_all: every mesh -- the
// GI mode's shadow rays do not skip refractive meshes, AccelerationStructure.cpp:66-71)
__device__ __forceinline__ v16f plan_leaf_of(const float4 *boxes, uint32_t k) {
    return *(kv16p)((kfp)(const float *)boxes + PLAN_LEAF_DWORDS * (size_t)k);
}

// Leaf cursor of the plan kernels: bits 0..23 = the next entry of the leaf, bits 24..30 = entries that follow it (compact
// leaf links, KernelArgs::pnodes); NONE = not inside a leaf.
__device__ __forceinline__ uint32_t leaf_cursor_entry(uint32_t c) { return c & 0x00FFFFFFu; }
__device__ __forceinline__ uint32_t leaf_cursor_next(uint32_t c) { return (c >> 24) ? c + 1u - (1u << 24) : NONE; }

// ---------------------------------------------------------------------------------------------------------------- shadow
// the shadow rays [first, first + total) of the queue; `cursor` hands them out
constexpr uint32_t SHADOW_NODE_REPEAT = 2;  // node steps per loop trip (measured: 2 beats 1, 3 and 4)
template <uint32_t WORDS>  // 32-bit words of a ray's mesh mask: 2 (the plan proper: <= 64 leaves, <= 64 meshes) or 8 (the wide plan)
__device__ __forceinline__ void shadow_plan_walks(const KernelArgs &A, const uint32_t first, const uint32_t total, uint32_t *cursor) {
    __shared__ uint32_t root_of_bit[32 * WORDS];  // shadow order -> root node of the mesh's tree
    const bool every_mesh = A.f->use_gi != 0;     // (uniform over the launch)
    const float4 *leaf_table = every_mesh ? A.s->plan_boxes_all : A.s->plan_boxes;
    {
        const uint32_t bits = every_mesh ? A.s->plan_shadow_bits_all : A.s->plan_shadow_bits;
        const uint32_t *order = every_mesh ? A.s->plan_shadow_mesh_all : A.s->plan_shadow_mesh;
        if (threadIdx.x < 32u * WORDS) root_of_bit[threadIdx.x] = threadIdx.x < bits ? A.s->meshes[order[threadIdx.x]].root : END;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const char *nodes_b = reinterpret_cast<const char *>(A.s->pnodes);
    const char *ptris_b = reinterpret_cast<const char *>(A.s->ptris);

    Ray R;
    uint32_t wn = END, we = NONE;   // next mesh-tree node, leaf cursor
    uint32_t todo[WORDS];           // meshes still to walk, bits of the shadow order
#pragma unroll
    for (uint32_t w = 0; w < WORDS; w++) todo[w] = 0;
    float light_dist = 0;
    uint32_t nbox = 0, ntri = 0, nplan = 0;
    int state = ST_FETCH;
    uint32_t r = 0, steps = 0;
    for (;;) {
        // ---- refill: free lanes wait until no more than `bundle` lanes still walk, then fetch together (the plan below
        //      costs the wave the same for one new ray as for 64)
        const uint32_t bundle = A.bundle;
        if (__ballot(state == ST_FETCH) && (bundle >= 64u || (uint32_t)__popcll(__ballot(state == ST_TRAVERSE)) <= bundle)) {
            bool fresh = false;
            while (state == ST_FETCH) {
                r = wave_fetch(cursor, lane);
                if (r >= total) { state = ST_DONE; break; }
                r += first;
                const float4 q0 = A.f->s_shadowq[2 * (size_t)r], q1 = A.f->s_shadowq[2 * (size_t)r + 1];
                if (__float_as_uint(q0.w) == SHADOW_SLOT_UNUSED) continue;  // a level-0 pixel without a diffuse hit
                // The light is behind the surface (angle = max(0, l . n) = 0, RayTracer.cpp:312): its contribution is
                // intensity / area * 0 = +-0 times the albedo, and adding +-0 to the light sum -- which starts at +0 and can
                // therefore never be -0 -- changes no bit of it (RayTracer.cpp:319-328).  Occluded or not, the pixel is the same:
                // no walk.  (A NaN factor -- the light AT the surface -- is not zero and is walked.)
                if (q1.w == 0.0f) { A.f->s_occluded[r] = 0; continue; }
                R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;  // already normalised once; shadow rays skip shootRay (RayTracer.cpp:313-317)
                ray_prepare(R);
                if (R.parmask != 0) {  // BoundingBox.h:90-93 needs the general test: the wave-per-ray kernel has it
                    if (evict_ray(A.f->s_sheavy, A.f->s_heavy_cap, A.f->s_counts + SC_SHEAVY, r, lane)) continue;
                    A.f->s_counts[SC_OVERFLOW] = 1;
                    continue;
                }
                light_dist = q0.w;
                fresh = true;
                steps = 0;
                state = ST_TRAVERSE;
            }
            if (fresh) {
                // the plan: which meshes does this ray have to walk?  (k is wave-uniform: scalar loads, no gathers)
                uint32_t acc[WORDS];
#pragma unroll
                for (uint32_t w = 0; w < WORDS; w++) acc[w] = 0;
                if constexpr (WORDS == 2) {
                    for (uint32_t k = 0; k < A.s->plan_leaves; k++) {
                        const v16f L = plan_leaf_of(leaf_table, k);
                        const bool hit = slab_test_no_parallel(R, L[0], L[1], L[2], L[4], L[5], L[6]);
                        acc[0] |= hit ? __float_as_uint(L[8]) : 0u;
                        acc[1] |= hit ? __float_as_uint(L[9]) : 0u;
                    }
                    if (A.exec_count) nplan += A.s->plan_leaves;
                } else {
                    for (uint32_t g = 0; g < A.s->plan_group_count; g++) {
                        const v8f G = plan_group(A, g);
                        const bool gh = slab_test_no_parallel(R, G[0], G[1], G[2], G[4], G[5], G[6]);
                        const uint32_t k0 = __builtin_amdgcn_readfirstlane(__float_as_uint(G[3]));
                        const uint32_t n = __ballot(gh) ? __builtin_amdgcn_readfirstlane(__float_as_uint(G[7])) : 0u;
                        for (uint32_t k = k0; k < k0 + n; k++) {
                            const v16f L = plan_leaf_of(leaf_table, k);
                            const bool hit = slab_test_no_parallel(R, L[0], L[1], L[2], L[4], L[5], L[6]);
#pragma unroll
                            for (uint32_t w = 0; w < WORDS; w++) acc[w] |= hit ? __float_as_uint(L[8 + w]) : 0u;
                        }
                        if (A.exec_count) nplan += 1u + n;
                    }
                }
#pragma unroll
                for (uint32_t w = 0; w < WORDS; w++) todo[w] = acc[w];
                wn = END; we = NONE;
            }
        }
        if (!__ballot(state != ST_DONE)) break;
        if (state == ST_TRAVERSE) {
            bool done = false, occluded = false;
            // one mesh-tree node (KDTree.cpp:53-74, BoundingBox.h:85-108), branch-free, the step to the next mesh folded in;
            // false: no mesh is left, the walk is complete
            auto node_step = [&]() -> bool {
                if (wn == END) {
                    uint32_t at = WORDS, word = 0;  // the first word with a mesh left
#pragma unroll
                    for (uint32_t w = WORDS; w-- > 0;)
                        if (todo[w]) { at = w; word = todo[w]; }
                    if (at == WORDS) return false;
#pragma unroll
                    for (uint32_t w = 0; w < WORDS; w++)
                        if (w == at) todo[w] = word & (word - 1u);
                    wn = root_of_bit[32u * at + (uint32_t)__builtin_ctz(word)];
                }
                const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(wn << 5));
                const float4 q0 = N[0], q1 = N[1];
                if (A.exec_count) nbox++;
                const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
                const bool hit = slab_test_no_parallel(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
                const bool leaf = is_leaf_link(link);
                we = (hit && leaf) ? (link & ~LEAF) : NONE;
                wn = (hit && !leaf) ? link : miss;
                return true;
            };
            for (int it = 0; it < 64; ++it) {  // (a lane whose walk ends waits for the end of this block of trips)
                if (we != NONE) {
                    steps++;
                    // ---- one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
                    const float4 *T = reinterpret_cast<const float4 *>(ptris_b + (size_t)(leaf_cursor_entry(we) * 48u));
                    const float4 a = T[0], b = T[1], c = T[2];
                    if (A.exec_count) ntri++;
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
                    we = leaf_cursor_next(we);
                    if (ok && t < INFINITY && shadow_hit_occludes(R, px, py, pz, light_dist)) { occluded = true; done = true; break; }
                } else {
                    steps++;
                    if (!node_step()) { done = true; break; }
                }
                // Further node steps in the same trip for the lanes that are (still) between leaves: a ray takes about two
                // node steps per triangle step, and a node step costs the wave a third of a triangle step, so a trip of
                // [triangle, node, node] keeps more lanes busy in the expensive block than [triangle, node] does.
                for (uint32_t rep = 1; rep < SHADOW_NODE_REPEAT; rep++) {
                    if (we == NONE) {
                        steps++;
                        if (!node_step()) { done = true; break; }
                    }
                }
                if (done) break;
            }
            if (done) {
                A.f->s_occluded[r] = occluded ? 1 : 0;
                state = ST_FETCH;
            } else if (steps >= A.step_budget) {
                // a long walk goes to the wave-per-ray kernel, which restarts it (the list full: keep walking here)
                if (evict_ray(A.f->s_sheavy, A.f->s_heavy_cap, A.f->s_counts + SC_SHEAVY, r, lane)) state = ST_FETCH;
                steps = 0;
            }
        }
    }
    exec_counters_flush(A, nbox, ntri, lane, nplan);
}

// pass 0: the level-0 shadow rays (queue slots below the split mark); pass 1: all the deeper levels' at once
template <uint32_t pass>  // (a template parameter so that the passes are separate kernels in a profile)
__global__ __launch_bounds__(BLOCK) void stream_trace_shadow_plan(const KernelArgs A) {
    const uint32_t split = A.f->s_counts[SC_SHADOW_SPLIT];
    shadow_plan_walks<2>(A, pass == 0 ? 0u : split, pass == 0 ? split : A.f->s_counts[SC_SHADOW] - split,
                         A.f->s_counts + (pass == 0 ? SC_SHADOW_FETCH : SC_SHADOW_FETCH2));
}
template <uint32_t pass>  // the same over the wide plan (SceneArgs::plan_wide)
__global__ __launch_bounds__(BLOCK) void stream_trace_shadow_plan_wide(const KernelArgs A) {
    const uint32_t split = A.f->s_counts[SC_SHADOW_SPLIT];
    shadow_plan_walks<8>(A, pass == 0 ? 0u : split, pass == 0 ? split : A.f->s_counts[SC_SHADOW] - split,
                         A.f->s_counts + (pass == 0 ? SC_SHADOW_FETCH : SC_SHADOW_FETCH2));
}

// ---------------------------------------------------------------------------------------------------------- closest hit
// The meshes a closest-hit ray has to walk, in the reference's order, as a byte list per lane in LDS: word w of thread t at
// list[w * BLOCK + t] holds entries 4w .. 4w+3 (conflict-free: consecutive lanes, consecutive words).
struct PlanList {
    uint32_t *words;   // this thread's column
    uint32_t count;    // entries written
    uint32_t next;     // next entry to read
};

// Appends, for every top-level leaf the ray reaches, the meshes it lists that the ray has not met before -- in leaf visit
// order and list order, i.e. in the order the reference collects their hits (KDTree.cpp:132-155).  Wave-uniform loops:
// leaf boxes and entries come through scalar loads.  WIDE: leaf groups (above), up to 256 meshes, and a list that may not
// hold them all (SceneArgs::plan_list_words is capped): false when this ray's list is full -- the ray goes to heavy_trace_closest.
template <bool WIDE>
__device__ __forceinline__ bool plan_closest_meshes(const KernelArgs &A, const Ray &R, PlanList &PL, uint32_t &nplan) {
    const ku32p entries = (ku32p)A.s->leaf_meshes;
    constexpr uint32_t WORDS = WIDE ? 8u : 2u;
    uint32_t seen[WORDS];
#pragma unroll
    for (uint32_t w = 0; w < WORDS; w++) seen[w] = 0;
    uint32_t acc = 0, cnt = 0;
    bool fits = true;
    const uint32_t cap = A.s->plan_list_words * 4u;
    auto leaf = [&](const uint32_t k) {
        const v16f L = plan_leaf(A, k);
        const bool hit = slab_test_no_parallel(R, L[0], L[1], L[2], L[4], L[5], L[6]);
        // (no `continue` on the wave-uniform ballot: DESIGN.md "compiler notes")
        const uint32_t first = __builtin_amdgcn_readfirstlane(__float_as_uint(L[3]));
        const uint32_t n = __ballot(hit) ? __builtin_amdgcn_readfirstlane(__float_as_uint(L[7])) : 0u;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t m = entries[first + j] & ~LAST;  // uniform, < 32 WORDS (plan_ok / plan_wide)
            const uint32_t bit = 1u << (m & 31u);
            bool fresh = false;
#pragma unroll
            for (uint32_t w = 0; w < WORDS; w++)
                if ((m >> 5) == w) { fresh = hit && !(seen[w] & bit); seen[w] |= fresh ? bit : 0u; }
            if (WIDE && fresh && cnt >= cap) { fits = false; fresh = false; }
            if (fresh) {
                acc |= m << (8u * (cnt & 3u));
                cnt++;
                if ((cnt & 3u) == 0u) { PL.words[((cnt >> 2) - 1u) * BLOCK] = acc; acc = 0; }
            }
        }
    };
    if constexpr (!WIDE) {
        for (uint32_t k = 0; k < A.s->plan_leaves; k++) leaf(k);
        if (A.exec_count) nplan += A.s->plan_leaves;
    } else {
        for (uint32_t g = 0; g < A.s->plan_group_count; g++) {
            const v8f G = plan_group(A, g);
            const bool gh = slab_test_no_parallel(R, G[0], G[1], G[2], G[4], G[5], G[6]);
            const uint32_t k0 = __builtin_amdgcn_readfirstlane(__float_as_uint(G[3]));
            const uint32_t n = __ballot(gh) ? __builtin_amdgcn_readfirstlane(__float_as_uint(G[7])) : 0u;
            for (uint32_t k = k0; k < k0 + n; k++) leaf(k);
            if (A.exec_count) nplan += 1u + n;
        }
    }
    if (cnt & 3u) PL.words[(cnt >> 2) * BLOCK] = acc;
    PL.count = cnt;
    PL.next = 0;
    return fits;
}

__device__ __forceinline__ uint32_t plan_list_pop(PlanList &PL) {
    const uint32_t i = PL.next++;
    return (PL.words[(i >> 2) * BLOCK] >> (8u * (i & 3u))) & 255u;
}

// Every ray of recursion level `gen`, one per lane, in the reference's own order: plan, walk of the listed meshes over the binary
// threaded nodes, material dispatch.  WIDE: the wide plan (SceneArgs::plan_wide).  Rays with a parallel axis and walks longer than the step budget go to heavy_trace_closest.
// (Measured and removed, DESIGN.md section 7: 4-wide quad nodes with an LDS stack -- the same frame time; the same nearest slot
// first with exact distance pruning on loose boxes -- grazing rays, the long walks, have nothing to prune before they hit.)
// SPLIT: the walk only -- the closest hit goes to s_hits_all[r] and stream_shade_all (kernel_stream.h) shades the level in a
// launch of its own, every lane busy; the GI mode's levels run this way (the sample code inlined into the walk loop costs it
// 30 vector registers and 60 spilled scalar ones).
template <bool WIDE, bool GI, bool SPLIT = false>
__device__ __forceinline__ void shade_plan_level(const KernelArgs &A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);  // ahead of the bulk shadow pass's waves on this SIMD
    extern __shared__ uint32_t plan_lds[];  // mesh lists: A.s->plan_list_words x BLOCK
    constexpr uint32_t MESHES = WIDE ? 256u : 64u;
    __shared__ uint32_t s_tree_root[MESHES];  // per mesh
    if (threadIdx.x < MESHES) s_tree_root[threadIdx.x] = threadIdx.x < A.s->top_meshes ? A.s->meshes[threadIdx.x].root : END;
    __syncthreads();
    PlanList PL;
    PL.words = plan_lds + threadIdx.x;
    PL.count = 0; PL.next = 0;
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const float4 *in_q = A.f->s_rayq[gen & 1u];
    const bool primary = gen == 0;
    if (stream_level_is_whole_heavy(A, gen, count)) return;
    const char *nodes_b = reinterpret_cast<const char *>(A.s->pnodes);
    const char *ptris_b = reinterpret_cast<const char *>(A.s->ptris);

    Ray R;
    uint32_t wn = END, we = NONE;   // next mesh-tree node (END: none), leaf cursor
    uint32_t mesh = NONE, mtri = 0, btri = 0, bmesh = 0;
    float mmin = INFINITY, mt = 0, tmin = INFINITY, bt = 0;
    bool mhave = false, have = false;
    uint32_t nbox = 0, ntri = 0, nplan = 0;
    int state = ST_FETCH;
    uint32_t r = 0, steps = 0;
    for (;;) {
        if (__ballot(state == ST_FETCH) && (A.bundle >= 64u || (uint32_t)__popcll(__ballot(state == ST_TRAVERSE)) <= A.bundle)) {
            bool fresh = false;
            while (state == ST_FETCH) {
                r = wave_fetch(A.f->s_counts + SC_FETCH + gen, lane);
                if (r >= count) { state = ST_DONE; break; }
                if (gen == 0) {
                    const Level0Ray P = level0_decode<GI>(A, r);
                    if (!P.covered) {
                        reinterpret_cast<uint32_t *>(A.f->s_nodes + 2 * (size_t)r)[3] = TN_SKIP;
                        level0_release_shadow_slots(A, r);
                        if (SPLIT) A.f->s_hits_all[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(3u));
                        continue;
                    }
                    level0_ray<GI>(A, P, R);
                } else {
                    const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
                    R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                    R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
                    normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                    ray_prepare(R);
                }
                if (R.parmask != 0) {
                    if (SPLIT) A.f->s_hits_all[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(2u));
                    if (evict_ray(A.f->s_heavy, A.f->s_heavy_cap, A.f->s_counts + SC_HEAVY + gen, r, lane)) continue;
                    A.f->s_counts[SC_OVERFLOW] = 1;  // cannot walk it here: let the fallback redo the frame
                    continue;
                }
                fresh = true;
                steps = 0;
                state = ST_TRAVERSE;
            }
            if (fresh) {
                const bool fits = plan_closest_meshes<WIDE>(A, R, PL, nplan);
                wn = END; we = NONE; mesh = NONE;
                mhave = false; mmin = INFINITY; mt = 0; mtri = 0;
                have = false; tmin = INFINITY; bt = 0; btri = 0; bmesh = 0;
                if (WIDE && !fits) {  // more meshes than this lane's list holds: the wave-per-ray kernel needs no list
                    if (SPLIT) A.f->s_hits_all[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(2u));
                    if (evict_ray(A.f->s_heavy, A.f->s_heavy_cap, A.f->s_counts + SC_HEAVY + gen, r, lane)) state = ST_FETCH;
                    else { A.f->s_counts[SC_OVERFLOW] = 1; state = ST_FETCH; }
                }
            }
        }
        if (!__ballot(state != ST_DONE)) break;
        if (state == ST_TRAVERSE) {
            bool done = false;
            for (int it = 0; it < 64; ++it) {
                steps++;
                if (we != NONE) {
                    // ---- one triangle of the current leaf (Ray.cpp:9-31, Triangle.cpp:37-57), branch-free
                    const uint32_t entry = leaf_cursor_entry(we);
                    const float4 *T = reinterpret_cast<const float4 *>(ptris_b + (size_t)(entry * 48u));
                    const float4 a = T[0], b = T[1], c = T[2];
                    if (A.exec_count) ntri++;
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
                    const bool ok = !(primary && nd >= 0) && !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
                    // `closest = hits[0]; min = inf; for h: if (h.d < min) { min = h.d; closest = h; }` (KDTree.cpp:75-86)
                    const bool less = ok && (t < mmin);
                    const bool take = less || (ok && !mhave);
                    mt = take ? t : mt;
                    mtri = take ? entry : mtri;  // the leaf ENTRY: its triangle is looked up once, when the ray is shaded
                    mmin = less ? t : mmin;
                    mhave = mhave || ok;
                    we = leaf_cursor_next(we);
                } else {
                    if (wn == END) {
                        // ---- a mesh ended (scene-level rule, KDTree.cpp:156-167), the next one begins
                        if (mesh != NONE && mhave) {
                            if (!have) { have = true; bt = mt; btri = mtri; bmesh = mesh; }
                            if (mt < tmin) { tmin = mt; bt = mt; btri = mtri; bmesh = mesh; }
                        }
                        if (PL.next >= PL.count) { done = true; break; }
                        mesh = plan_list_pop(PL);
                        wn = s_tree_root[mesh];
                        mhave = false;
                        mmin = INFINITY;
                    }
                    // ---- one mesh-tree node (KDTree.cpp:53-74, BoundingBox.h:85-108), branch-free
                    const float4 *N = reinterpret_cast<const float4 *>(nodes_b + (size_t)(uint32_t)(wn << 5));
                    const float4 q0 = N[0], q1 = N[1];
                    if (A.exec_count) nbox++;
                    const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
                    const bool hit = slab_test_no_parallel(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
                    const bool leaf = is_leaf_link(link);
                    we = (hit && leaf) ? (link & ~LEAF) : NONE;
                    wn = (hit && !leaf) ? link : miss;
                }
            }
            if (done) {
                if (have) btri = A.s->leaf_tris[btri] & ~LAST;  // leaf entry -> triangle
                if constexpr (SPLIT) A.f->s_hits_all[r] = make_float4(bt, __uint_as_float(btri), __uint_as_float(bmesh), __uint_as_float(have ? 1u : 0u));
                else shade_and_emit<false, GI>(A, gen, r, node_base, child_base, R, have, bt, btri, bmesh, nullptr, lane);
                state = ST_FETCH;
            } else if (steps >= A.step_budget) {
                if (evict_ray(A.f->s_heavy, A.f->s_heavy_cap, A.f->s_counts + SC_HEAVY + gen, r, lane)) {
                    if (SPLIT) A.f->s_hits_all[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(2u));
                    state = ST_FETCH;
                }
                steps = 0;
            }
        }
    }
    exec_counters_flush(A, nbox, ntri, lane, nplan);
}

__global__ __launch_bounds__(BLOCK, 4) void stream_trace_shade_plan(const KernelArgs A, const uint32_t gen) { shade_plan_level<false, false>(A, gen); }
__global__ __launch_bounds__(BLOCK) void stream_trace_shade_plan_wide(const KernelArgs A, const uint32_t gen) { shade_plan_level<true, false>(A, gen); }
// The GI / multi-sample mode's builds (RayTracer.cpp:90-104, 331-354; jittered level-0 samples) are walk-only (SPLIT): stream_shade_all<true>
// shades the level -- gi_samples child rays per diffuse hit.  (Measured for the plain frame too: HW14 344 vs 356, HW12 421 vs 432 Mpixels/s -- not used there.)
__global__ __launch_bounds__(BLOCK) void stream_trace_plan_gi(const KernelArgs A, const uint32_t gen) { shade_plan_level<false, true, true>(A, gen); }
__global__ __launch_bounds__(BLOCK) void stream_trace_plan_wide_gi(const KernelArgs A, const uint32_t gen) { shade_plan_level<true, true, true>(A, gen); }
