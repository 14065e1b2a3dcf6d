// kernel_bfs.h -- level-synchronous (breadth-first) tree walks for the rays a recursion level hands over.
//
// A recursion level >= 1 holds 20,000 - 60,000 secondary rays, and the nine levels of a frame depend on one another: the
// levels are the frame's critical path.  Walked depth-first -- one ray per lane, per group of lanes or per wave -- a ray is a
// chain of 30 - 80 dependent memory round trips, and a level lasts as long as its rays' chains however many lanes a ray is
// given (measured, DESIGN.md section 7: one ray per wave 67 rays/us, one per 8 or 16 lanes the same, all of it waiting).
// Here the walk is turned inside out: the unit of work is not a ray but one (ray, chunk of the tree) PAIR, and a level's walks
// run as a handful of passes over queues of such pairs in HBM:
//
//   bfs_plan   one ray per 16 lanes: the ray is prepared once (origin, direction, 1 / direction: 48 bytes per ray), tested
//              against the top-level leaves (the plan, kernel_plan.h), and every mesh it has to walk -- in the reference's
//              visit order, each at its first occurrence -- becomes an item {ray, mesh, its visit rank, the top chunk}.
//   bfs_pass   x (levels of the deepest leaf sequence + 1): one item per 16 lanes.  A box item tests the 16 entries of its
//              chunk with the reference's exact arithmetic (BoundingBox.h:85-108) and appends one item per passing entry --
//              the chunk below it, or, for a leaf, its triangle list; a triangle item tests up to 16 triangles (Ray.cpp:9-31,
//              Triangle.cpp:37-57) and folds accepted hits into the ray's result with one atomic.
//   bfs_shade  one ray per lane: the material dispatch (kernel_stream.h: shade_and_emit) on the winning triangle.
//
// Every pass is a plain data-parallel kernel: items are independent, lanes are full, nothing waits for anything but its own
// two loads, and the dependent depth of a recursion level is the depth of the hierarchy (5 passes for 224,000 leaves at a
// fan-out of 16), not the length of a ray's walk.
//
// Why the result is the reference's, bit for bit.  The reference collects a mesh's hits in visit order and keeps the first
// one, replaced only by a STRICTLY smaller distance (KDTree.cpp:75-86); across meshes the same rule again (KDTree.cpp:156-167).
// That is: the hit with the smallest finite distance, ties going to the earliest in (mesh visit order, leaf order, position in
// the leaf); and when no hit has a finite distance, the first accepted hit in that order (a NaN or infinite distance is
// accepted by Ray.cpp:19 but never wins `d < min`).  Both are order-independent reductions -- a minimum over
// (distance, rank, leaf entry) and a minimum over (rank, leaf entry) -- so the items may run in any order: one 64-bit and
// one 32-bit atomicMin per accepted hit.  Leaf entries of a mesh ascend in visit order (the flattener lays the leaf lists
// out that way), and the rank is the mesh's position in the ray's own visit order.  Which leaves are visited is decided by
// the leaves' own boxes (nested boxes + monotone slab test: kernel_heavy.h), the union boxes above them only skip leaves
// that would fail anyway.  The winning distance itself is not carried through the atomics: bfs_shade repeats the one
// triangle test, the same instructions on the same operands.
#pragma once

#include "kernel_common.h"
#include "kernel_stream.h"
#include "kernel_walk.h"

constexpr int BFS_F = 16;             // fan-out of the leaf-sequence hierarchy = lanes per item
constexpr int BFS_LOGF = 4;
constexpr int BFS_LEVELS = 6;         // 16^6 = 2^24 leaves
constexpr int BFS_PASSES = BFS_LEVELS + 1;  // box passes + the triangle pass
constexpr int BFS_ROWS = BFS_PASSES + 1;    // counter rows per set: what the plan appended, what every pass appended
constexpr int BFS_SHARDS = 32;        // every queue is BFS_SHARDS independent regions, each with a counter on a cache line of its own
constexpr int BFS_COUNTER_STRIDE = 32;  // words between two counters (128 bytes: atomics on one line are served one by one)
constexpr int BFS_SETS = MAX_GENERATIONS + 2;  // one set of pass counters per recursion level + the two shadow parts
constexpr int BFS_PLAN_LEAVES = 256, BFS_PLAN_ENTRIES = 1024, BFS_PLAN_MESHES = 256;  // what bfs_plan's LDS tables hold
constexpr uint32_t BFS_KIND_TRI = 6u, BFS_KIND_TINY = 7u;  // item `lvl` values beyond the box levels 0..5

// item = {list entry k, mesh | rank << 16 | lvl << 24 | (valid entries - 1) << 27, a, b}
//   box item:  a = chunk index at `lvl`, b = index of the chunk's first entry in KernelArgs::gbox
//   triangle item (lvl = BFS_KIND_TRI): a = first leaf entry, b = triangles
//   single-leaf mesh (lvl = BFS_KIND_TINY): the same with b = triangles | index of the leaf's box in gbox << 8
__device__ __forceinline__ uint32_t bfs_meshword(uint32_t mesh, uint32_t rank, uint32_t lvl, uint32_t n) {
    return mesh | (rank << 16) | (lvl << 24) | ((n - 1u) << 27);
}

__device__ __forceinline__ uint32_t *bfs_counter(const KernelArgs &A, uint32_t set, uint32_t pass, uint32_t shard) {
    return A.b_counts + ((size_t)(set * BFS_ROWS + pass) * BFS_SHARDS + shard) * BFS_COUNTER_STRIDE;
}

// Wave-aggregated append of one item per emitting lane to queue `q` (shard `shard`): consecutive slots, one atomic per wave.
__device__ __forceinline__ void bfs_append(const KernelArgs &A, uint4 *q, uint32_t *counter, uint32_t shard, bool emit, const uint4 item, uint32_t lane) {
    const unsigned long long m = __ballot(emit);
    if (!m) return;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (emit) {
        if (slot < A.b_shard_cap) q[(size_t)shard * A.b_shard_cap + slot] = item;
        else A.s_counts[SC_OVERFLOW] = 1;  // the fallback redoes the frame; the queues grow (crt_device.hip: adapt_queue_sizing)
    }
}

// A prepared ray: {origin, parallel-axis mask} {direction, light distance} {1 / direction, ray id}
struct BfsRay { Ray R; float light_dist; uint32_t r; };
__device__ __forceinline__ BfsRay bfs_load_ray(const KernelArgs &A, uint32_t k) {
    const float4 p0 = A.b_prep[3 * (size_t)k], p1 = A.b_prep[3 * (size_t)k + 1], p2 = A.b_prep[3 * (size_t)k + 2];
    BfsRay B;
    B.R.ox = p0.x; B.R.oy = p0.y; B.R.oz = p0.z; B.R.parmask = __float_as_uint(p0.w);
    B.R.dx = p1.x; B.R.dy = p1.y; B.R.dz = p1.z; B.light_dist = p1.w;
    B.R.ix = p2.x; B.R.iy = p2.y; B.R.iz = p2.z; B.r = __float_as_uint(p2.w);
    return B;
}

// ---------------------------------------------------------------------------------------------------------------- plan
// List entries [first, total) of the level's (or the shadow list's) hand-over list: prepare the ray, find its meshes.
template <bool SHADOW>
__device__ __forceinline__ void bfs_plan_rays(const KernelArgs &A, const uint32_t gen, const bool whole, const uint32_t first, const uint32_t total,
                                              const uint32_t set) {
    constexpr int G = BFS_F, NG = BLOCK / G;
    __shared__ uint32_t s_leaf_first[BFS_PLAN_LEAVES], s_leaf_count[BFS_PLAN_LEAVES];
    __shared__ uint16_t s_entries[BFS_PLAN_ENTRIES];
    __shared__ lds_v4u s_mesh[BFS_PLAN_MESHES];
    __shared__ uint32_t s_hitw[NG][BFS_PLAN_LEAVES / 32], s_seenw[NG][BFS_PLAN_MESHES / 32];
    __shared__ uint32_t s_found[NG][G];  // the meshes of the current batch: mesh | rank << 16
    for (uint32_t i = threadIdx.x; i < A.plan_leaves; i += BLOCK) {
        const float4 b0 = A.plan_boxes[4 * (size_t)i], b1 = A.plan_boxes[4 * (size_t)i + 1];
        s_leaf_first[i] = __float_as_uint(b0.w);
        s_leaf_count[i] = __float_as_uint(b1.w);
    }
    for (uint32_t i = threadIdx.x; i < A.top_leaf_entries; i += BLOCK) s_entries[i] = (uint16_t)(A.leaf_meshes[i] & ~LAST);
    for (uint32_t i = threadIdx.x; i < A.top_meshes; i += BLOCK) { const uint4 m = A.gmesh[i]; s_mesh[i] = lds_v4u{m.x, m.y, m.z, m.w}; }
    __syncthreads();
    if (A.s_counts[SC_OVERFLOW]) return;
    const uint32_t lane = threadIdx.x & 63u, gl = threadIdx.x & (G - 1u), g = threadIdx.x / G;
    const uint32_t gbase = lane & ~(G - 1u);
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t shard = wave_id % BFS_SHARDS;
    uint32_t *const out_counter = bfs_counter(A, set, 0u, shard);
    uint4 *const out_q = A.b_items[0];
    const float4 *in_q = A.s_rayq[gen & 1u];
    const uint32_t n_groups = gridDim.x * NG;
    // (every group of the wave runs the same number of trips: the appends below are wave-wide)
    const uint32_t span = total > first ? total - first : 0u;
    const uint32_t trips = (span + n_groups - 1u) / n_groups;
    for (uint32_t trip = 0; trip < trips; trip++) {
        const uint32_t k = first + trip * n_groups + blockIdx.x * NG + g;
        const bool live = k < total;
        Ray R;
        R.ox = R.oy = R.oz = 0; R.dx = R.dy = 0; R.dz = 1; R.ix = R.iy = R.iz = 0; R.parmask = 0;
        float light_dist = 0;
        uint32_t r = 0;
        if (live) {
            if (SHADOW) {
                r = A.s_sheavy[k];
                const float4 q0 = A.s_shadowq[2 * (size_t)r], q1 = A.s_shadowq[2 * (size_t)r + 1];
                R.ox = q0.x; R.oy = q0.y; R.oz = q0.z; light_dist = q0.w;
                R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;  // normalised once; shadow rays skip shootRay (RayTracer.cpp:313-317)
                ray_prepare(R);
            } else {
                r = whole ? k : A.s_heavy[k];
                if (gen == 0) {
                    const WorkItem wi = A.items[r >> 6];
                    const uint32_t sub = r & 63u;
                    primary_ray(A, (wi.tile % A.tiles_x) * TILE + (sub & 7u), (wi.tile / A.tiles_x) * TILE + (sub >> 3), R);
                } else {
                    const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
                    R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                    R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
                    normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                    ray_prepare(R);
                }
            }
            if (gl == 0) {
                A.b_prep[3 * (size_t)k] = make_float4(R.ox, R.oy, R.oz, __uint_as_float(R.parmask));
                A.b_prep[3 * (size_t)k + 1] = make_float4(R.dx, R.dy, R.dz, light_dist);
                A.b_prep[3 * (size_t)k + 2] = make_float4(R.ix, R.iy, R.iz, __uint_as_float(r));
                if (SHADOW) A.s_occluded[r] = 0;
                else { A.b_best[k] = ~0ull; A.b_firstnf[k] = ~0u; }
            }
        }
        // ---- the top-level leaves the ray passes, 16 at a time
        if (gl < (uint32_t)(BFS_PLAN_LEAVES / 32)) s_hitw[g][gl] = 0u;
        if (gl < (uint32_t)(BFS_PLAN_MESHES / 32)) s_seenw[g][gl] = 0u;
        for (uint32_t base = 0; base < A.plan_leaves; base += G) {  // (wave-uniform trip count)
            const uint32_t idx = base + gl;
            bool hit = false;
            if (live && idx < A.plan_leaves) {
                const float4 b0 = A.plan_boxes[4 * (size_t)idx], b1 = A.plan_boxes[4 * (size_t)idx + 1];
                hit = slab_test(R, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
            }
            const uint32_t bits = (uint32_t)(__ballot(hit) >> gbase) & 0xFFFFu;
            if (gl == 0 && bits) s_hitw[g][base >> 5] |= bits << (base & 31u);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- its meshes: leaves in visit order, entries in list order, every mesh at its first occurrence (KDTree.cpp:132-155;
        //      kernel_common.h: mesh_walk_is_repeat), in batches of 16 per group; lane j of the group then emits the j-th
        uint32_t cur_word = 0, cur_bits = s_hitw[g][0], ent = 0, ent_end = 0, rank = 0;
        bool more = live;
        while (__ballot(more)) {
            uint32_t n_found = 0;
            while (more && n_found < (uint32_t)G) {
                if (ent < ent_end) {
                    const uint32_t mi = s_entries[ent++];
                    const lds_v4u m = s_mesh[mi];
                    if (SHADOW && (m[2] & 1u) && !A.use_gi) continue;  // AccelerationStructure.cpp:67-71
                    const uint32_t bit = 1u << (mi & 31u);
                    const uint32_t w = s_seenw[g][mi >> 5];
                    if (w & bit) continue;
                    s_seenw[g][mi >> 5] = w | bit;
                    if (m[1] == 0u) continue;  // a mesh without leaves
                    s_found[g][n_found++] = mi | (rank << 16);
                    rank++;
                    continue;
                }
                while (cur_bits == 0u && cur_word + 1u < (uint32_t)(BFS_PLAN_LEAVES / 32)) cur_bits = s_hitw[g][++cur_word];
                if (cur_bits == 0u) { more = false; break; }
                const uint32_t leaf = cur_word * 32u + (uint32_t)__builtin_ctz(cur_bits);
                cur_bits &= cur_bits - 1u;
                ent = s_leaf_first[leaf];
                ent_end = ent + s_leaf_count[leaf];
            }
            __builtin_amdgcn_wave_barrier();
            const bool emit = gl < n_found;
            uint4 item = make_uint4(0, 0, 0, 0);
            if (emit) {
                const uint32_t f = s_found[g][gl], mi = f & 0xFFFFu, rk = f >> 16;
                const lds_v4u m = s_mesh[mi];
                if (m[3] != NONE) {  // a single leaf: its box and its triangles are tested in one step (BFS_KIND_TINY)
                    const uint32_t cnt = ((m[3] >> 24) & 127u) + 1u;
                    item = make_uint4(k, bfs_meshword(mi, rk, BFS_KIND_TINY, 1u), leaf_cursor_entry(m[3]), cnt | (m[0] << 8));
                } else {
                    // the top level of its leaf sequence: level 0 = the leaves, level l = unions of 16 entries of level l - 1
                    uint32_t fst = m[0], cnt = m[1], l = 0;
                    while (cnt > (uint32_t)BFS_F) { fst += cnt; cnt = (cnt + BFS_F - 1u) >> BFS_LOGF; l++; }
                    item = make_uint4(k, bfs_meshword(mi, rk, l, cnt), 0u, fst);
                }
            }
            bfs_append(A, out_q, out_counter, shard, emit, item, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <bool SHADOW>
__global__ __launch_bounds__(BLOCK) void bfs_plan(const KernelArgs A, const uint32_t gen_or_part) {
    if (!SHADOW && A.wave_prio) __builtin_amdgcn_s_setprio(3);  // ahead of the bulk shadow pass's waves on this SIMD
    if (SHADOW) {
        uint32_t total = A.s_counts[SC_SHEAVY], split = A.s_counts[SC_SHEAVY_SPLIT];
        if (total > A.s_heavy_cap) total = A.s_heavy_cap;
        if (split > total) split = total;
        bfs_plan_rays<true>(A, 0u, false, gen_or_part == 0 ? 0u : split, gen_or_part == 0 ? split : total, (uint32_t)MAX_GENERATIONS + gen_or_part);
    } else {
        const uint32_t gen = gen_or_part;
        const uint32_t count = stream_level_count(A, gen);
        const bool whole = stream_level_is_whole_heavy(A, gen, count);  // every ray of the level: entry k is ray k
        uint32_t total = whole ? count : A.s_counts[SC_HEAVY + gen];
        if (total > A.s_heavy_cap) total = A.s_heavy_cap;
        if (A.force_whole && !whole && gen > 0) {  // no per-lane launch and more rays than the lists hold: the fallback redoes the frame
            if (threadIdx.x == 0 && blockIdx.x == 0) A.s_counts[SC_OVERFLOW] = 1;
            total = 0;
        }
        bfs_plan_rays<false>(A, gen, whole, 0u, total, gen);
    }
}

// ---------------------------------------------------------------------------------------------------------------- passes
// Pass `pass` of counter set `set`: consumes queue pass & 1 (what pass - 1, or the plan, appended), appends to queue (pass + 1) & 1.
// A pass is a chain of three dependent memory operations per item -- the item, then its ray and its boxes or triangles, then
// the slot for what it appends -- and the chip's wave slots are all taken, so what decides its duration is how many items a
// wave has in flight: every 16-lane group takes IPG consecutive items per trip, requests everything they need at once,
// and the whole wave reserves its output slots with one atomic.
template <bool SHADOW, int IPG>
__global__ __launch_bounds__(BLOCK) void bfs_pass(const KernelArgs A, const uint32_t set, const uint32_t pass, const uint32_t primary_flag) {
    if (!SHADOW && A.wave_prio) __builtin_amdgcn_s_setprio(3);
    constexpr int G = BFS_F, NG = BLOCK / G;
    if (A.s_counts[SC_OVERFLOW]) return;
    const bool primary = primary_flag != 0;
    const uint32_t lane = threadIdx.x & 63u, gl = threadIdx.x & (G - 1u), g = threadIdx.x / G;
    // workgroup b serves shard b % BFS_SHARDS, batch b / BFS_SHARDS, then every (gridDim.x / BFS_SHARDS)-th batch of it
    const uint32_t in_shard = blockIdx.x % BFS_SHARDS;
    const uint32_t wgs_per_shard = gridDim.x / BFS_SHARDS;  // (the grid is a multiple of BFS_SHARDS)
    uint32_t n_items = *bfs_counter(A, set, pass, in_shard);
    if (n_items > A.b_shard_cap) n_items = A.b_shard_cap;
    const uint4 *in_q = A.b_items[pass & 1u] + (size_t)in_shard * A.b_shard_cap;
    uint4 *const out_q = A.b_items[(pass + 1u) & 1u];
    const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t out_shard = wave_id % BFS_SHARDS;
    uint32_t *const out_counter = bfs_counter(A, set, pass + 1u, out_shard);
    const char *const gbox_b = reinterpret_cast<const char *>(A.gbox);
    const char *const ptris_b = reinterpret_cast<const char *>(A.ptris);
    uint32_t nbox = 0, ntri = 0;
    for (uint32_t batch = blockIdx.x / BFS_SHARDS; batch * (NG * IPG) < n_items; batch += wgs_per_shard) {  // (workgroup-uniform trip count)
        // ---- the items
        uint4 it[IPG];
        bool live[IPG];
#pragma unroll
        for (int j = 0; j < IPG; j++) {
            const uint32_t i = batch * (NG * IPG) + g * IPG + (uint32_t)j;
            live[j] = i < n_items;
            it[j] = in_q[live[j] ? i : 0u];
        }
        // ---- their rays, and per lane one box (32 bytes) or one triangle (48 bytes): unconditional loads, addresses clamped
        float4 p0[IPG], p1[IPG], p2[IPG], q0[IPG], q1[IPG], q2[IPG];
        uint2 lv[IPG];
        bool mine[IPG];   // this lane has an entry / a triangle of item j
#pragma unroll
        for (int j = 0; j < IPG; j++) {
            const uint32_t k = it[j].x, mesh = it[j].y & 0xFFFFu, lvl = (it[j].y >> 24) & 7u, n = (it[j].y >> 27) + 1u;
            p0[j] = A.b_prep[3 * (size_t)k]; p1[j] = A.b_prep[3 * (size_t)k + 1]; p2[j] = A.b_prep[3 * (size_t)k + 2];
            const bool box = lvl < BFS_KIND_TRI;
            const uint32_t cnt = lvl == BFS_KIND_TINY ? (it[j].w & 255u) : it[j].w;   // (triangle items)
            mine[j] = live[j] && (box ? gl < n : gl < cnt);
            const char *addr = box ? gbox_b + (size_t)(it[j].w + gl) * 32u : ptris_b + (size_t)(it[j].z + gl) * 48u;
            if (!mine[j]) addr = gbox_b;
            q0[j] = *reinterpret_cast<const float4 *>(addr);
            q1[j] = *reinterpret_cast<const float4 *>(addr + 16);
            q2[j] = *reinterpret_cast<const float4 *>(addr + (box ? 16 : 32));   // (a box has no third quarter)
            lv[j] = A.glevels[mesh * BFS_LEVELS + ((box && lvl > 0u) ? lvl - 1u : 0u)];  // {first entry, entries} of the level below
        }
        // ---- tests
        bool emit[IPG];
        uint4 child[IPG];
#pragma unroll
        for (int j = 0; j < IPG; j++) {
            const uint32_t k = it[j].x, mesh = it[j].y & 0xFFFFu, rank = (it[j].y >> 16) & 255u, lvl = (it[j].y >> 24) & 7u;
            Ray R;
            R.ox = p0[j].x; R.oy = p0[j].y; R.oz = p0[j].z; R.parmask = __float_as_uint(p0[j].w);
            R.dx = p1[j].x; R.dy = p1[j].y; R.dz = p1[j].z;
            R.ix = p2[j].x; R.iy = p2[j].y; R.iz = p2[j].z;
            const float light_dist = p1[j].w;
            const uint32_t r = __float_as_uint(p2[j].w);
            emit[j] = false;
            child[j] = make_uint4(0, 0, 0, 0);
            if (lvl < BFS_KIND_TRI) {
                // a chunk of boxes: entry `gl` of it on this lane (BoundingBox.h:85-108)
                if (mine[j]) {
                    if (A.exec_count) nbox++;
                    if (slab_test(R, q0[j].x, q0[j].y, q0[j].z, q1[j].x, q1[j].y, q1[j].z)) {
                        emit[j] = true;
                        if (lvl == 0u) child[j] = make_uint4(k, bfs_meshword(mesh, rank, BFS_KIND_TRI, 1u), __float_as_uint(q0[j].w), __float_as_uint(q1[j].w));
                        else {
                            const uint32_t c = (it[j].z << BFS_LOGF) + gl;                 // this entry's chunk on the level below
                            const uint32_t left = lv[j].y - (c << BFS_LOGF);
                            child[j] = make_uint4(k, bfs_meshword(mesh, rank, lvl - 1u, left < (uint32_t)BFS_F ? left : (uint32_t)BFS_F), c, lv[j].x + (c << BFS_LOGF));
                        }
                    }
                }
            } else if (live[j]) {
                // the triangles of a leaf (Ray.cpp:9-31, Triangle.cpp:37-57), the first 16 from the loads above
                const uint32_t begin = it[j].z, cnt = lvl == BFS_KIND_TINY ? (it[j].w & 255u) : it[j].w;
                bool boxhit = true;
                if (lvl == BFS_KIND_TINY) {  // a single-leaf mesh: the leaf's own box decides whether its triangles count
                    const float4 b0 = A.gbox[2 * (size_t)(it[j].w >> 8)], b1 = A.gbox[2 * (size_t)(it[j].w >> 8) + 1];
                    boxhit = slab_test(R, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
                    if (A.exec_count && gl == 0) nbox++;
                }
                for (uint32_t off = 0; boxhit && off < cnt; off += G) {
                    if (off + gl < cnt) {
                        const uint32_t e = begin + off + gl;
                        float4 a = q0[j], b = q1[j], c = q2[j];
                        if (off) { a = A.ptris[3 * (size_t)e]; b = A.ptris[3 * (size_t)e + 1]; c = A.ptris[3 * (size_t)e + 2]; }  // (a leaf of more than 16 triangles)
                        const float plane = -dot3(a.x, a.y, a.z, a.w, b.w, c.w);  // distanceToPlane, Ray.cpp:17 (checked against the host's at crt_create)
                        float t = 0;
                        if (A.exec_count) ntri++;
                        if (triangle_test(R, primary, a, b, c, plane, t)) {
                            if (SHADOW) {
                                // exact: an accepted hit with a finite distance within the light's ends the question (kernel_walk.h)
                                if (t < INFINITY && shadow_hit_occludes(R, R.ox + R.dx * t, R.oy + R.dy * t, R.oz + R.dz * t, light_dist)) A.s_occluded[r] = 1;
                            } else {
                                const uint32_t order = (rank << 24) | e;  // visit rank of the mesh, leaf entry (ascending in visit order)
                                if (t < INFINITY) {
                                    // (t >= 0 here: its bits order like the floats; -0 orders with +0 -- the tie then goes by `order`, as in the reference)
                                    const uint32_t tb = t == 0.0f ? 0u : __float_as_uint(t);
                                    atomicMin(&A.b_best[k], ((unsigned long long)tb << 32) | order);
                                } else atomicMin(&A.b_firstnf[k], order);
                            }
                        }
                    }
                }
            }
        }
        // ---- what they append: consecutive slots, one atomic for the wave
        unsigned long long em[IPG];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < IPG; j++) { em[j] = __ballot(emit[j]); total += (uint32_t)__popcll(em[j]); }
        if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(out_counter, total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int j = 0; j < IPG; j++) {
                const uint32_t slot = base + (uint32_t)__popcll(em[j] & ((1ull << lane) - 1ull));
                if (emit[j]) {
                    if (slot < A.b_shard_cap) out_q[(size_t)out_shard * A.b_shard_cap + slot] = child[j];
                    else A.s_counts[SC_OVERFLOW] = 1;  // the fallback redoes the frame; the queues grow (crt_device.hip: adapt_queue_sizing)
                }
                base += (uint32_t)__popcll(em[j]);
            }
        }
    }
    exec_counters_flush(A, nbox, ntri, lane);
}

// ---------------------------------------------------------------------------------------------------------------- shade
// Material dispatch for the rays of level `gen` walked above (kernel_stream.h: stream_shade_evicted for this path).
__global__ __launch_bounds__(BLOCK) void bfs_shade(const KernelArgs A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63u;
    if (A.s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    const bool whole = stream_level_is_whole_heavy(A, gen, count);
    uint32_t total = whole ? count : A.s_counts[SC_HEAVY + gen];
    if (total > A.s_heavy_cap) total = A.s_heavy_cap;
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const bool primary = gen == 0;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < total; k += gridDim.x * BLOCK) {
        const BfsRay B = bfs_load_ray(A, k);
        const unsigned long long best = A.b_best[k];
        const uint32_t nf = A.b_firstnf[k];
        const bool have = best != ~0ull || nf != ~0u;
        const uint32_t entry = (best != ~0ull ? (uint32_t)best : nf) & 0x00FFFFFFu;  // (bits 24..31: the mesh's visit rank)
        uint32_t tri = 0, mesh = 0;
        float t = 0;
        if (have) {
            tri = A.leaf_tris[entry] & ~LAST;
            mesh = A.tri_mesh[tri];
            // the winning distance: the same test on the same operands as the pass that accepted it
            // (a 32-bit byte offset: ROCm 7.2's gfx950 backend turns `zext(x & 0xFFFFFF) * 48` in 64 bits into v_mad_u64_u32 on
            // the UNMASKED x -- the 24-bit multiply's licence to ignore the upper bits carried over to an instruction that reads them)
            const float4 *T = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A.ptris) + entry * 48u);
            const float4 a = T[0], b = T[1], c = T[2];
            const float plane = -dot3(a.x, a.y, a.z, a.w, b.w, c.w);
            (void)triangle_test(B.R, primary, a, b, c, plane, t);
        }
        if (gen == 0) A.s_hits[k] = make_float4(t, __uint_as_float(tri), __uint_as_float(mesh), __uint_as_float(have ? 1u : 0u));  // DEBUG
        shade_and_emit<false>(A, gen, B.r, node_base, child_base, B.R, have, t, tri, mesh, nullptr, lane);
    }
}
