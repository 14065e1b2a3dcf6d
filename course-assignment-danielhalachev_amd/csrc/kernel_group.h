// kernel_group.h -- group_trace: one ray per GROUP of G lanes (G = 8 or 16), U boxes / triangles per lane and step.
//
// A per-lane walk (kernel_plan.h) is a chain of dependent loads, ~80 of them for a secondary ray of the benchmark frame and
// thousands for the longest ones: fine for the millions of primary and shadow rays, which hide each other's latency, but a
// recursion level of 20,000-60,000 secondary rays is all latency, and the nine levels of a frame depend on one another.
// Round 2 walked those rays one per WAVE (64 boxes / 64 triangles per instruction): ~30 dependent round trips per ray, but
// only as many rays in flight as the chip has wave slots, leaves of <= 8 triangles on 64 lanes, and every piece of per-ray
// state in scalar registers (122 spilled).  Here a ray gets G lanes:
//
//  * The walk runs over the LEAF SEQUENCE of a mesh tree, as before: within one of the reference's trees a child box is its
//    parent's box with one face moved inwards (BoundingBox.h:60-69) and the slab test (BoundingBox.h:85-108) is monotone under
//    that nesting, so "the reference visits this leaf" == "the leaf's own box passes the test" (crt_create verifies the
//    nesting).  Above the leaves sit union boxes of F = G x U consecutive entries per level; a ray that passes a box passes
//    any box containing it, so the union boxes only skip leaves that would fail anyway.
//  * One step of a group = F boxes of one chunk (lane j, slot u: entry u*G + j), or the triangles of up to U leaves (slot u:
//    one leaf, lane j its j-th triangle), tested with the reference's exact arithmetic.  The closest hit follows the
//    reference's rule -- first collected hit, replaced only by a strictly smaller distance, in list order (KDTree.cpp:75-86)
//    -- through a group-wide minimum on DPP (quad_perm / row_half_mirror / row_mirror: no LDS, no readlane) and the position
//    of the first lane that holds it.
//  * Per-ray state lives in vector registers (the same value in the G lanes of a group); the depth-first position in the
//    F-ary hierarchy (one F-bit mask of pending entries per level) and the leaves of the current chunk wait in a few hundred
//    bytes of LDS per group.  The groups of a wave are at different places of different walks: every trip of the loop issues
//    the loads of ALL groups first -- boxes or triangles, the same registers -- and tests afterwards, so a trip costs one
//    memory round trip whatever mixture of steps it holds.
//  * The top-level tree is a plan (kernel_plan.h): its leaves in visit order, tested F at a time like any other chunk; the
//    meshes they list are walked in the reference's order, each at its first occurrence (kernel_common.h:
//    mesh_walk_is_repeat).  Tables of up to 256 leaves / 1024 entries / 256 meshes sit in LDS.
//  * A mesh whose tree is a single leaf (a room's walls and floor) is tested in ONE step: its box and its triangles are
//    requested together.
#pragma once

#include "kernel_common.h"
#include "kernel_stream.h"
#include "kernel_walk.h"

constexpr int GROUP_LEVELS = 8;                                       // F^8 leaves at least 2^24 for F >= 8
constexpr int GPLAN_LEAVES = 256, GPLAN_ENTRIES = 1024, GPLAN_MESHES = 256;  // what the LDS tables hold (crt_create: KernelArgs::group_ok)

// per mesh: {first entry of its leaf sequence in KernelArgs::gbox, number of leaves, crt_mesh::flags,
//            single-leaf meshes: LEAF | (triangles - 1) << 24 | first leaf entry, else NONE}
typedef uint32_t gmesh_rec __attribute__((ext_vector_type(4)));

template <int G>
struct Grp {
    static_assert(G == 8 || G == 16, "a group is half a DPP row or a whole one");
    // bits of the calling lane's group in a wave ballot (bit j = lane j of the group)
    static __device__ __forceinline__ uint32_t ballot(bool p, uint32_t gbase) {
        const unsigned long long b = __ballot(p);
        return (uint32_t)(b >> gbase) & ((1u << G) - 1u);
    }
    template <int CTRL>
    static __device__ __forceinline__ float dpp(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
    }
    // minimum over the group, in every lane of it (no NaNs reach it: callers pass +inf for "no value")
    static __device__ __forceinline__ float min_all(float v) {
        v = fminf(v, dpp<0xB1>(v));   // quad_perm [1,0,3,2]
        v = fminf(v, dpp<0x4E>(v));   // quad_perm [2,3,0,1]
        v = fminf(v, dpp<0x141>(v));  // row_half_mirror: lane i <-> 7 - i of its half row
        if (G == 16) v = fminf(v, dpp<0x140>(v));  // row_mirror: lane i <-> 15 - i
        return v;
    }
    // the value lane `w` of the group holds (w: the same in every lane of the group)
    static __device__ __forceinline__ uint32_t from_lane(uint32_t v, uint32_t gbase, uint32_t w) {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((gbase + w) << 2), (int)v);
    }
};

enum : int { GP_FETCH = 0, GP_PLAN, GP_NEXT, GP_BOX, GP_TRI, GP_TINY, GP_POP, GP_DONE };

// per-group scratch in LDS (words)
template <int F>
struct GroupScratch {
    uint32_t leaf_begin[F], leaf_count[F];   // the leaves of the current level-0 chunk that the ray passes
    uint32_t lvl_mask[GROUP_LEVELS];          // pending entries of the chunk on the path at each level
    uint32_t lvl_first[GROUP_LEVELS], lvl_count[GROUP_LEVELS];  // the current mesh's levels in KernelArgs::gbox
    uint32_t hitw[GPLAN_LEAVES / 32];         // top-level leaves the ray passes
    uint32_t seenw[GPLAN_MESHES / 32];        // meshes already walked for this ray
};

// closest hits of list entries [0, total) of recursion level `gen` (SHADOW: occlusion flags of shadow-list entries
// [first, total)); `cursor` hands the entries out
template <int G, int U, bool SHADOW>
__device__ __forceinline__ void group_walks(const KernelArgs &A, const uint32_t gen, const bool whole, const uint32_t first,
                                            const uint32_t total, uint32_t *cursor) {
    constexpr int F = G * U, NG = BLOCK / G;
    constexpr int LOGF = F == 8 ? 3 : (F == 16 ? 4 : 5);
    static_assert(F == 8 || F == 16 || F == 32, "fan-out");
    __shared__ uint32_t s_leaf_first[GPLAN_LEAVES], s_leaf_count[GPLAN_LEAVES];
    __shared__ uint16_t s_entries[GPLAN_ENTRIES];
    __shared__ gmesh_rec s_mesh[GPLAN_MESHES];
    __shared__ GroupScratch<F> s_grp[NG];
    for (uint32_t i = threadIdx.x; i < A.plan_leaves; i += BLOCK) {
        s_leaf_first[i] = __float_as_uint(A.plan_boxes[4 * (size_t)i].w);
        s_leaf_count[i] = __float_as_uint(A.plan_boxes[4 * (size_t)i + 1].w);
    }
    for (uint32_t i = threadIdx.x; i < A.top_leaf_entries; i += BLOCK) s_entries[i] = (uint16_t)(A.leaf_meshes[i] & ~LAST);
    for (uint32_t i = threadIdx.x; i < A.top_meshes; i += BLOCK) s_mesh[i] = A.gmesh[i];
    __syncthreads();
    if (A.s_counts[SC_OVERFLOW]) return;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gl = lane & (G - 1u), gbase = lane & ~(G - 1u);
    GroupScratch<F> &S = s_grp[threadIdx.x / G];
    const bool primary = !SHADOW && gen == 0;
    const float4 *in_q = A.s_rayq[gen & 1u];

    int phase = GP_FETCH;
    uint32_t k = 0, r = 0;
    Ray R;
    R.ox = R.oy = R.oz = R.dx = R.dy = R.dz = R.ix = R.iy = R.iz = 0; R.parmask = 0;
    float light_dist = 0;
    uint32_t pchunk = 0, cur_word = 0, cur_bits = 0, ent = 0, ent_end = 0;
    uint32_t mesh = NONE, lvl = 0, top_lvl = 0, chunk = 0, lfirst = 0, lcount = 0, leafmask = 0, leaf_off = 0, tiny_link = NONE;
    bool mhave = false, have = false, occluded = false;
    float mmin = INFINITY, mt = 0, tmin = INFINITY, bt = 0;
    uint32_t mtri = 0, btri = 0, bmesh = 0;
    uint32_t guard = 0, nbox = 0, ntri = 0, n_steps = 0, n_meshes = 0;

    for (;;) {
        // ------------------------------------------------------------------ bookkeeping: registers and LDS only
        if (phase == GP_POP) {
            // the next pending entry on the path, deepest level first; none left: the mesh is finished
            for (;;) {
                const uint32_t m = S.lvl_mask[lvl];
                if (m) {
                    const uint32_t b = (uint32_t)__builtin_ctz(m);
                    S.lvl_mask[lvl] = m & (m - 1u);
                    chunk = (chunk << LOGF) + b;
                    lvl--;
                    lfirst = S.lvl_first[lvl]; lcount = S.lvl_count[lvl];
                    phase = GP_BOX;
                    break;
                }
                if (lvl == top_lvl) { phase = GP_NEXT; break; }
                lvl++;
                chunk >>= LOGF;
            }
        }
        if (phase == GP_NEXT) {
            // ---- the mesh just walked ends: scene-level rule (KDTree.cpp:156-167; AccelerationStructure.cpp:73-74)
            if (mesh != NONE && mhave) {
                if (SHADOW) {
                    const float px = R.ox + R.dx * mt, py = R.oy + R.dy * mt, pz = R.oz + R.dz * mt;
                    if (len3(px - R.ox, py - R.oy, pz - R.oz) <= light_dist) occluded = true;
                } else {
                    if (!have) { have = true; bt = mt; btri = mtri; bmesh = mesh; }
                    if (mt < tmin) { tmin = mt; bt = mt; btri = mtri; bmesh = mesh; }
                }
            }
            mesh = NONE;
            // ---- the next mesh: leaves in visit order, entries in list order, every mesh at its first occurrence
            bool more = !(SHADOW && occluded) && guard != 0;
            while (more) {
                if (ent < ent_end) {
                    const uint32_t mi = s_entries[ent++];
                    const gmesh_rec m = s_mesh[mi];
                    if (SHADOW && (m[2] & 1u) && !A.use_gi) continue;  // AccelerationStructure.cpp:67-71
                    const uint32_t bit = 1u << (mi & 31u);
                    const uint32_t w = S.seenw[mi >> 5];
                    if (w & bit) continue;
                    S.seenw[mi >> 5] = w | bit;
                    if (m[1] == 0u) continue;  // a mesh without leaves
                    mesh = mi;
                    mhave = false; mmin = INFINITY; mt = 0; mtri = 0;
                    n_meshes++;
                    tiny_link = m[3];
                    if (tiny_link != NONE) { lfirst = m[0]; phase = GP_TINY; break; }
                    // its levels: level 0 = the leaves, level l = unions of F entries of level l - 1, until <= F are left
                    uint32_t fst = m[0], cnt = m[1], l = 0;
                    for (;;) {
                        S.lvl_first[l] = fst; S.lvl_count[l] = cnt;
                        if (cnt <= (uint32_t)F) break;
                        fst += cnt; cnt = (cnt + F - 1u) >> LOGF; l++;
                    }
                    top_lvl = l; lvl = l; chunk = 0; lfirst = fst; lcount = cnt;
                    phase = GP_BOX;
                    break;
                }
                while (cur_bits == 0u && cur_word + 1u < (uint32_t)(GPLAN_LEAVES / 32)) cur_bits = S.hitw[++cur_word];
                if (cur_bits == 0u) { more = false; break; }
                const uint32_t leaf = cur_word * 32u + (uint32_t)__builtin_ctz(cur_bits);
                cur_bits &= cur_bits - 1u;
                ent = s_leaf_first[leaf];
                ent_end = ent + s_leaf_count[leaf];
            }
            if (!more) {
                // ---- the ray is finished
                if (gl == 0) {
                    if (SHADOW) A.s_occluded[r] = occluded ? 1 : 0;
                    else {
                        const uint32_t tri = have ? (A.leaf_tris[btri] & ~LAST) : 0u;  // leaf entry -> triangle
                        A.s_hits[k] = make_float4(bt, __uint_as_float(tri), __uint_as_float(bmesh), __uint_as_float(have ? 1u : 0u));
                    }
                    if (guard == 0) { A.s_counts[SC_GUARD] = 1; A.s_counts[SC_OVERFLOW] = 1; }  // bound hit: let the fallback redo the frame
                    if (A.exec_count) {
                        atomicAdd(&A.exec_counters[0], (unsigned long long)nbox); atomicAdd(&A.exec_counters[1], (unsigned long long)ntri);
                        uint32_t *diag = A.s_counts + SC_HEAVY_DIAG + (SHADOW ? 8 : 0);  // tools/stream_stats.py: what a walk consists of
                        atomicAdd(diag + 0, 1u); atomicAdd(diag + 1, n_steps); atomicAdd(diag + 3, n_meshes);
                    }
                }
                phase = GP_FETCH;
            }
        }
        // ------------------------------------------------------------------ a new ray
        if (__ballot(phase == GP_FETCH)) {
            if (phase == GP_FETCH) {
                const unsigned long long need = __ballot(gl == 0);  // one request per group
                uint32_t base = 0;
                if (lane == (uint32_t)(__ffsll((long long)need) - 1)) base = atomicAdd(cursor, (uint32_t)__popcll(need));
                base = (uint32_t)__shfl((int)base, __ffsll((long long)need) - 1);
                k = first + base + (uint32_t)__popcll(need & ((1ull << gbase) - 1ull));
                if (k >= total) phase = GP_DONE;
                else {
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
                    have = false; occluded = false; tmin = INFINITY; bt = 0; btri = 0; bmesh = 0;
                    mesh = NONE; mhave = false;
                    guard = 1u << 16;
                    nbox = 0; ntri = 0; n_steps = 0; n_meshes = 0;
                    if (gl < (uint32_t)(GPLAN_LEAVES / 32)) S.hitw[gl] = 0u;
                    if (gl < (uint32_t)(GPLAN_MESHES / 32)) S.seenw[gl] = 0u;
                    pchunk = 0;
                    phase = GP_PLAN;
                }
            }
        }
        if (!__ballot(phase != GP_DONE)) break;
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------------ this trip's loads, all groups' at once
        const bool boxlike = phase == GP_BOX || phase == GP_PLAN, trilike = phase == GP_TRI || phase == GP_TINY;
        float4 q0[U], q1[U], q2[U], tb0, tb1;
        bool valid[U];
        uint32_t ebase[U];   // trilike: the leaf entry lane 0 of the slot tests
        tb0 = make_float4(0, 0, 0, 0); tb1 = tb0;
        uint32_t taken = 0;     // GP_TRI: leaves taken from leafmask in this step
        bool partial = false;   // GP_TRI: slot 0's leaf is longer than the group and continues in the next step
        if (phase != GP_DONE && guard) { guard--; n_steps++; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            q0[u] = make_float4(0, 0, 0, 0); q1[u] = q0[u]; q2[u] = q0[u];
            valid[u] = false; ebase[u] = 0;
            const float4 *p = nullptr;
            if (phase == GP_BOX) {
                const uint32_t idx = (chunk << LOGF) + (uint32_t)u * G + gl;
                valid[u] = idx < lcount;
                p = A.gbox + 2 * (size_t)(lfirst + idx);
            } else if (phase == GP_PLAN) {
                const uint32_t idx = pchunk * F + (uint32_t)u * G + gl;
                valid[u] = idx < A.plan_leaves;
                p = A.plan_boxes + 4 * (size_t)idx;
            } else if (phase == GP_TINY) {
                const uint32_t begin = leaf_cursor_entry(tiny_link), cnt = (tiny_link >> 24 & 127u) + 1u;
                ebase[u] = begin + (uint32_t)u * G;
                valid[u] = (uint32_t)u * G + gl < cnt;
                p = A.ptris + 3 * (size_t)(ebase[u] + gl);
            } else if (phase == GP_TRI) {
                // slot u: the u-th pending leaf of the chunk; a leaf longer than the group is taken G triangles at a time, alone
                uint32_t m = leafmask;
                for (int s = 0; s < u; s++) m &= m - 1u;
                if (m != 0u && !partial) {
                    const uint32_t b = (uint32_t)__builtin_ctz(m);
                    const uint32_t begin = S.leaf_begin[b], cnt = S.leaf_count[b], off = u == 0 ? leaf_off : 0u;
                    if (u == 0 || cnt <= (uint32_t)G) {
                        ebase[u] = begin + off;
                        valid[u] = off + gl < cnt;
                        p = A.ptris + 3 * (size_t)(ebase[u] + gl);
                        if (u == 0 && cnt - off > (uint32_t)G) partial = true;
                        else taken = (uint32_t)u + 1u;
                    }
                }
            }
            if (valid[u]) {
                q0[u] = p[0]; q1[u] = p[1];
                if (trilike) q2[u] = p[2];
            }
        }
        if (phase == GP_TINY) { tb0 = A.gbox[2 * (size_t)lfirst]; tb1 = A.gbox[2 * (size_t)lfirst + 1]; }

        // ------------------------------------------------------------------ boxes (BoundingBox.h:85-108)
        if (boxlike) {
            uint32_t bits = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool hit = valid[u] && slab_test(R, q0[u].x, q0[u].y, q0[u].z, q1[u].x, q1[u].y, q1[u].z);
                bits |= Grp<G>::ballot(hit, gbase) << (u * G);
                if (A.exec_count && phase == GP_BOX) nbox += (uint32_t)__popc(Grp<G>::ballot(valid[u], gbase));
                if (phase == GP_BOX && lvl == 0u && hit) {
                    S.leaf_begin[u * G + gl] = __float_as_uint(q0[u].w);
                    S.leaf_count[u * G + gl] = __float_as_uint(q1[u].w);
                }
            }
            if (phase == GP_PLAN) {
                // F divides 32: the chunk's bits lie inside one word
                if (gl == 0 && bits) S.hitw[(pchunk * F) >> 5] |= bits << ((pchunk * F) & 31u);
                pchunk++;
                if (pchunk * F >= A.plan_leaves) {
                    __builtin_amdgcn_wave_barrier();
                    cur_word = 0; cur_bits = S.hitw[0]; ent = 0; ent_end = 0;
                    phase = GP_NEXT;
                }
            } else if (lvl == 0u) {
                leafmask = bits; leaf_off = 0;
                phase = bits ? GP_TRI : GP_POP;
            } else if (bits) {
                // into the first passing entry at once; the others wait in this level's mask
                S.lvl_mask[lvl] = bits & (bits - 1u);
                chunk = (chunk << LOGF) + (uint32_t)__builtin_ctz(bits);
                lvl--;
                lfirst = S.lvl_first[lvl]; lcount = S.lvl_count[lvl];
            } else {
                S.lvl_mask[lvl] = 0u;
                phase = GP_POP;
            }
        }
        // ------------------------------------------------------------------ triangles (Ray.cpp:9-31, Triangle.cpp:37-57)
        if (trilike) {
            float t[U];
            bool acc[U];
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float4 a = q0[u], b = q1[u], c = q2[u];
                const float plane = -dot3(a.x, a.y, a.z, a.w, b.w, c.w);  // distanceToPlane, Ray.cpp:17 (checked against the host's at crt_create)
                t[u] = 0;
                acc[u] = valid[u] && triangle_test(R, primary, a, b, c, plane, t[u]);
                if (A.exec_count) ntri += (uint32_t)__popc(Grp<G>::ballot(valid[u], gbase));
                any = any || acc[u];
            }
            if (phase == GP_TINY) {  // the leaf's own box decides whether its triangles count
                const bool boxhit = slab_test(R, tb0.x, tb0.y, tb0.z, tb1.x, tb1.y, tb1.z);
                if (A.exec_count) nbox++;
#pragma unroll
                for (int u = 0; u < U; u++) acc[u] = acc[u] && boxhit;
                any = any && boxhit;
            }
            if (__ballot(any)) {
                uint32_t am[U];
                uint32_t any_bits = 0;
#pragma unroll
                for (int u = 0; u < U; u++) { am[u] = Grp<G>::ballot(acc[u], gbase); any_bits |= am[u]; }
                if (any_bits) {
                    if (SHADOW) {
                        bool occ = false;
#pragma unroll
                        for (int u = 0; u < U; u++)
                            occ = occ || (acc[u] && t[u] < INFINITY &&
                                          shadow_hit_occludes(R, R.ox + R.dx * t[u], R.oy + R.dy * t[u], R.oz + R.dz * t[u], light_dist));
                        if (Grp<G>::ballot(occ, gbase)) occluded = true;  // exact early exit (kernel_walk.h)
                    }
                    if (!mhave) {  // `closest = intersections[0]`: the first accepted hit in list order, whatever its distance
                        uint32_t fu = 0, fm = am[0];
#pragma unroll
                        for (int u = U - 1; u >= 1; u--) { bool earlier_none = true; for (int s = 0; s < u; s++) earlier_none = earlier_none && am[s] == 0u; if (earlier_none && am[u]) { fu = (uint32_t)u; fm = am[u]; } }
                        const uint32_t w = (uint32_t)__builtin_ctz(fm);
                        float tf = t[0];
#pragma unroll
                        for (int u = 1; u < U; u++) tf = fu == (uint32_t)u ? t[u] : tf;
                        uint32_t eb = ebase[0];
#pragma unroll
                        for (int u = 1; u < U; u++) eb = fu == (uint32_t)u ? ebase[u] : eb;
                        mhave = true;
                        mt = __uint_as_float(Grp<G>::from_lane(__float_as_uint(tf), gbase, w));
                        mtri = eb + w;
                    }
                    // +inf / NaN distances never win `d < min`; ties inside the step go to the earliest in list order
                    float key[U], kmin = INFINITY;
#pragma unroll
                    for (int u = 0; u < U; u++) { key[u] = (acc[u] && t[u] < INFINITY) ? t[u] : INFINITY; kmin = fminf(kmin, key[u]); }
                    const float mn = Grp<G>::min_all(kmin);
                    if (mn < mmin) {  // strictly smaller than everything collected before
                        uint32_t wu = 0, wm = 0;
#pragma unroll
                        for (int u = U - 1; u >= 0; u--) {
                            const uint32_t cm = Grp<G>::ballot(acc[u] && key[u] == mn, gbase);
                            if (cm) { wu = (uint32_t)u; wm = cm; }
                        }
                        const uint32_t w = (uint32_t)__builtin_ctz(wm);
                        float tw = t[0];
#pragma unroll
                        for (int u = 1; u < U; u++) tw = wu == (uint32_t)u ? t[u] : tw;
                        uint32_t eb = ebase[0];
#pragma unroll
                        for (int u = 1; u < U; u++) eb = wu == (uint32_t)u ? ebase[u] : eb;
                        mmin = mn;
                        mt = __uint_as_float(Grp<G>::from_lane(__float_as_uint(tw), gbase, w));
                        mtri = eb + w;
                    }
                }
            }
            if (phase == GP_TINY) phase = GP_NEXT;
            else {
                if (partial) leaf_off += G;
                else {
                    for (uint32_t s = 0; s < taken; s++) leafmask &= leafmask - 1u;
                    leaf_off = 0;
                }
                if (leafmask == 0u) phase = GP_POP;
            }
            if (SHADOW && occluded) phase = GP_NEXT;  // (GP_NEXT ends an occluded ray)
        }
        if (guard == 0 && phase != GP_DONE && phase != GP_FETCH) phase = GP_NEXT;  // bound hit: end the ray (the fallback redoes the frame)
        __builtin_amdgcn_wave_barrier();
    }
}

// closest hits of the rays stream_trace_shade(gen) handed over -- or of every ray of the level (stream_level_is_whole_heavy);
// results go to s_hits[k] for list entry k
template <int G, int U>
__global__ __launch_bounds__(BLOCK) void group_trace_closest(const KernelArgs A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);  // ahead of the bulk shadow pass's waves on this SIMD
    const uint32_t count = stream_level_count(A, gen);
    const bool whole = stream_level_is_whole_heavy(A, gen, count);  // every ray of the level: entry k is ray k
    uint32_t total = whole ? count : A.s_counts[SC_HEAVY + gen];
    if (total > A.s_heavy_cap) total = A.s_heavy_cap;
    if (A.force_whole && !whole && gen > 0) {  // no per-lane launch and more rays than s_hits holds: the fallback redoes the frame
        if (threadIdx.x == 0 && blockIdx.x == 0) A.s_counts[SC_OVERFLOW] = 1;
        total = 0;
    }
    group_walks<G, U, false>(A, gen, whole, 0u, total, A.s_counts + SC_HEAVY_FETCH + gen);
}

// part 0: the walks shadow pass 0 gave up (list entries below the SC_SHEAVY_SPLIT mark), on the side stream right
// after that pass; part 1: the rest, after the last pass.
template <int G, int U>
__global__ __launch_bounds__(BLOCK) void group_trace_shadow(const KernelArgs A, const uint32_t part) {
    uint32_t total = A.s_counts[SC_SHEAVY], split = A.s_counts[SC_SHEAVY_SPLIT];
    if (total > A.s_heavy_cap) total = A.s_heavy_cap;
    if (split > total) split = total;
    group_walks<G, U, true>(A, 0u, false, part == 0 ? 0u : split, part == 0 ? split : total,
                            A.s_counts + (part == 0 ? SC_SHEAVY_FETCH : SC_SHEAVY_FETCH2));
}
