// kernel_heavy.h -- heavy_trace: ONE ray per wavefront, 64 boxes / 64 triangles per instruction.
//
// A per-lane walk is a chain of dependent loads; a ray that crosses thousands of leaves holds its whole
// launch hostage (measured: 8,500 steps, 6 ms, on the benchmark frame).  The per-lane kernels therefore stop
// a walk after `step_budget` steps and queue the ray here, where the wave walks the tree for that single ray:
//
//  * Within one of the reference's trees a child box is its parent's box with one face moved inwards
//    (BoundingBox.h:60-69) and the slab test (BoundingBox.h:85-108) is monotone under rounding, so a ray
//    that passes a leaf's box passes every ancestor's: "the reference visits this leaf" == "the leaf's own
//    box passes the test".  (crt_create verifies the nesting; without it this kernel is not used.)
//  * So the walk can run over the LEAF SEQUENCE (leaves in the reference's visit order): 64 consecutive leaf
//    boxes are tested by the 64 lanes at once with the reference's exact arithmetic; above them sit union
//    boxes of 64 entries each (a ray that passes a box passes any box containing it, by the same
//    monotonicity), which only skip leaves that would fail anyway.
//  * The triangles of a passing leaf are tested one per lane; the winner is chosen with the reference's
//    rule -- first collected hit, replaced only by a strictly smaller distance, in list order
//    (KDTree.cpp:75-86) -- by a wave reduction keyed on (distance, position in the list).
#pragma once

#include "kernel_common.h"
#include "kernel_stream.h"
#include "kernel_walk.h"

struct HeavyState {   // all wave-uniform
    bool mhave;
    float mmin, mt;
    uint32_t mtri;
    uint32_t guard;   // remaining loop iterations for this ray: every loop of the walk is bounded by it
    bool stop;        // shadow walks: an accepted hit within the light's distance ends the walk (kernel_walk.h: shadow_hit_occludes)
    float light_dist;
    uint32_t nbox, ntri;  // tests executed for this ray (KernelArgs::exec_count)
    uint32_t n_chunks, n_batches, n_meshes;  // diagnostics (exec_count): 64-box chunk tests, 64-triangle batches, mesh trees entered
    __device__ __forceinline__ bool go() const { return guard != 0 && !stop; }
};

// Cross-lane helpers on the DPP / readlane paths (a ds_bpermute round trip per step is what the walk would
// otherwise wait on: every step of it is wave-uniform control flow around one reduction).
__device__ __forceinline__ float lane_value(float v, int lane_uniform) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane_uniform) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_uniform);
}

// Keys are distances of accepted hits: never negative (Ray.cpp:19 rejects t < 0), +inf for "none", and -0 is made +0 below -- so their
// order is the order of their bit patterns as unsigned integers, and an integer minimum folds into one DPP instruction per step
// (a float minimum costs five: the compiler canonicalises both operands of every fminf).
// minimum over the wave, returned uniformly (NaNs never reach it: callers pass +inf for "no value"); the result compares equal, as a
// float, to the key of every lane that holds the minimum.  Six DPP steps (row_shr 1 / 2 / 4 / 8, then row_bcast 15 into rows 1 and 3 and
// row_bcast 31 into rows 2 and 3: lane 63 ends with the wave's minimum) through the compiler's DPP builtin, which knows the wait states
// a DPP read needs after a VALU write and after a write of EXEC (round 3 had them as inline assembly, which the hazard recogniser cannot
// see into); a lane without a source (or outside the row mask) gets `old` = its own value.  EVERY lane of the wave must be active at the
// call: all call sites are in wave-uniform control flow (kernel_heavy.h's loops run on ballots).
__device__ __forceinline__ float wave_min(float key) {
    uint32_t v = __float_as_uint(key + 0.0f);  // (-0 -> +0: the one non-negative value whose pattern is out of order)
#define CRT_DPP_MIN(ctrl, rows) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false); v = o < v ? o : v; }
    CRT_DPP_MIN(0x111, 0xf) CRT_DPP_MIN(0x112, 0xf) CRT_DPP_MIN(0x114, 0xf) CRT_DPP_MIN(0x118, 0xf)   // row_shr:1, 2, 4, 8: lane 15 of every row has its row's minimum
    CRT_DPP_MIN(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    CRT_DPP_MIN(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef CRT_DPP_MIN
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)v, 63));
}

// One batch of up to 64 triangles (lane j tests list entry e when `valid`), lanes in list order.
template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_batch(const KernelArgs &A, const Ray &R, const bool primary, const bool valid,
                                            const size_t e, HeavyState &H) {
    const float4 a = A.s->ltris[4 * e + 0], b = A.s->ltris[4 * e + 1], c = A.s->ltris[4 * e + 2], d = A.s->ltris[4 * e + 3];
    float t = 0;
    if (EXEC) { H.ntri += (uint32_t)__popcll(__ballot(valid)); H.n_batches++; }
    const bool acc = valid && triangle_test(R, primary, a, b, c, d.x, t);
    const unsigned long long am = __ballot(acc);
    if (!am) return;
    if (SHADOW) {
        const bool occ = acc && t < INFINITY &&
                         shadow_hit_occludes(R, R.ox + R.dx * t, R.oy + R.dy * t, R.oz + R.dz * t, H.light_dist);
        if (__ballot(occ)) { H.stop = true; return; }
    }
    if (!H.mhave) {  // `closest = intersections[0]`: the first accepted hit in list order, whatever its distance
        const int first = __ffsll((long long)am) - 1;
        H.mhave = true;
        H.mt = lane_value(t, first);
        H.mtri = lane_value(__float_as_uint(d.y), first);
    }
    const float key = (acc && t < INFINITY) ? t : INFINITY;  // +inf / NaN distances never win `d < min`
    const float mn = wave_min(key);
    if (mn < H.mmin) {  // strictly smaller than everything collected before; ties inside the batch: earliest
        const unsigned long long wm = __ballot(acc && key == mn);
        const int w = __ffsll((long long)wm) - 1;
        H.mmin = mn;
        H.mt = lane_value(t, w);
        H.mtri = lane_value(__float_as_uint(d.y), w);
    }
}

// The triangles of all leaves of one 64-leaf chunk whose box the ray passes (`hit` lanes: leaf = lane,
// list [begin, begin+count)), concatenated in leaf order and tested 64 at a time.  The concatenation is laid
// over the lanes with wave-uniform arithmetic: `off` lanes of the batch under construction are taken.
template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_leaves(const KernelArgs &A, const Ray &R, const bool primary, const bool hit,
                                             const uint32_t begin, const uint32_t count, HeavyState &H, const uint32_t lane) {
    unsigned long long m = __ballot(hit);
    uint32_t off = 0, e = 0;
    bool valid = false;
    while (m && H.go()) {
        H.guard--;
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint32_t bg = lane_value(begin, k), cnt = lane_value(count, k);
        while (cnt && H.go()) {  // a leaf longer than the free lanes continues in the next batch
            const uint32_t take = cnt < 64u - off ? cnt : 64u - off;
            if (lane >= off && lane < off + take) { e = bg + (lane - off); valid = true; }
            off += take; bg += take; cnt -= take;
            if (off == 64u) {
                H.guard--;
                heavy_batch<SHADOW, EXEC>(A, R, primary, valid, (size_t)e, H);
                off = 0;
                valid = false;
            }
        }
    }
    if (off && H.go()) heavy_batch<SHADOW, EXEC>(A, R, primary, valid, (size_t)e, H);
}

struct ChunkBoxes { float4 b0, b1; bool valid; };

template <int LEVEL>
__device__ __forceinline__ ChunkBoxes heavy_chunk_load(const KernelArgs &A, const HeavyMesh &M, const uint32_t chunk, const uint32_t lane) {
    ChunkBoxes C;
    const uint32_t idx = chunk * 64u + lane;
    C.valid = idx < M.count[LEVEL];
    const size_t at = (size_t)M.first[LEVEL] + (C.valid ? idx : 0u);
    C.b0 = A.s->hbox[2 * at];
    C.b1 = A.s->hbox[2 * at + 1];
    return C;
}

template <int LEVEL, bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_chunk(const KernelArgs &A, const Ray &R, const bool primary, const HeavyMesh &M,
                                            const uint32_t chunk, const ChunkBoxes &C, HeavyState &H, const uint32_t lane) {
    if (EXEC) { H.nbox += (uint32_t)__popcll(__ballot(C.valid)); H.n_chunks++; }
    bool hit = C.valid && slab_test(R, C.b0.x, C.b0.y, C.b0.z, C.b1.x, C.b1.y, C.b1.z);
    if constexpr (LEVEL == 0) {
        if (__ballot(hit)) heavy_leaves<SHADOW, EXEC>(A, R, primary, hit, __float_as_uint(C.b0.w), __float_as_uint(C.b1.w), H, lane);
    } else {
        unsigned long long m = __ballot(hit);
        if (!m) return;
        while (m && H.go()) {
            H.guard--;
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const ChunkBoxes cur = heavy_chunk_load<LEVEL - 1>(A, M, chunk * 64u + (uint32_t)k, lane);
            heavy_chunk<LEVEL - 1, SHADOW, EXEC>(A, R, primary, M, chunk * 64u + (uint32_t)k, cur, H, lane);
        }
    }
}

// ---- single-leaf meshes, all at once ----------------------------------------------------------------------------------
// A room's walls and floor are meshes whose tree is one leaf.  Walked one after the other each costs the ray two dependent
// loads (its box, its triangles) -- a third of the dependent steps of a typical walk.  Their results do not depend on WHEN
// they are computed, only the order in which the scene-level rule sees them does: so at the start of a ray lane k tests the
// box of single-leaf mesh k (KernelArgs::tiny_at), the triangles of those that pass are tested in one or two batches, and the
// per-mesh results wait in lane k's registers until the top-level walk reaches mesh k, in the reference's order.
struct TinyResults {  // per lane k < tiny_count: the closest hit in single-leaf mesh k (mesh-level rule, KDTree.cpp:75-86)
    bool have;
    float mmin, mt;
    uint32_t mtri;
};

template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_tiny_batch(const KernelArgs &A, const Ray &R, const bool primary, const bool valid, const size_t e,
                                                 const uint32_t tag, unsigned long long tags, TinyResults &T, HeavyState &H, const uint32_t lane) {
    const float4 a = A.s->ltris[4 * e + 0], b = A.s->ltris[4 * e + 1], c = A.s->ltris[4 * e + 2], d = A.s->ltris[4 * e + 3];
    float t = 0;
    if (EXEC) { H.ntri += (uint32_t)__popcll(__ballot(valid)); H.n_batches++; }
    const bool acc = valid && triangle_test(R, primary, a, b, c, d.x, t);
    if (!__ballot(acc)) return;
    while (tags) {  // the meshes that have triangles in this batch, one masked reduction each (registers only)
        const int g = __ffsll((long long)tags) - 1;
        tags &= tags - 1;
        const bool mine = acc && tag == (uint32_t)g;
        const unsigned long long am = __ballot(mine);
        if (!am) continue;
        bool have = __builtin_amdgcn_readlane((int)T.have, g) != 0;
        float mmin = lane_value(T.mmin, g), mt = lane_value(T.mt, g);
        uint32_t mtri = lane_value(T.mtri, g);
        if (!have) {  // `closest = intersections[0]`
            const int first = __ffsll((long long)am) - 1;
            have = true;
            mt = lane_value(t, first);
            mtri = lane_value(__float_as_uint(d.y), first);
        }
        const float key = (mine && t < INFINITY) ? t : INFINITY;
        const float mn = wave_min(key);
        if (mn < mmin) {
            const int w = __ffsll((long long)__ballot(mine && key == mn)) - 1;
            mmin = mn;
            mt = lane_value(t, w);
            mtri = lane_value(__float_as_uint(d.y), w);
        }
        if (lane == (uint32_t)g) { T.have = have; T.mmin = mmin; T.mt = mt; T.mtri = mtri; }
    }
}

template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_tiny_meshes(const KernelArgs &A, const Ray &R, const bool primary, TinyResults &T, HeavyState &H,
                                                  const uint32_t lane) {
    T.have = false; T.mmin = INFINITY; T.mt = 0; T.mtri = 0;
    if (A.s->tiny_count == 0) return;
    const bool valid = lane < A.s->tiny_count;
    const size_t at = valid ? A.s->tiny_at[lane] : A.s->tiny_at[0];
    const uint32_t flags = valid ? A.s->tiny_flags[lane] : 0u;
    const float4 b0 = A.s->hbox[2 * at], b1 = A.s->hbox[2 * at + 1];
    if (EXEC) H.nbox += A.s->tiny_count;
    const bool hit = valid && !(SHADOW && (flags & 1u) && !A.f->use_gi) && slab_test(R, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z);
    unsigned long long m = __ballot(hit);
    if (!m) return;
    const uint32_t begin = __float_as_uint(b0.w), count = __float_as_uint(b1.w);
    uint32_t off = 0, e = 0, tag = 0;
    bool tvalid = false;
    unsigned long long tags = 0;
    while (m && H.guard) {
        H.guard--;
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint32_t bg = lane_value(begin, k), cnt = lane_value(count, k);
        while (cnt && H.guard) {
            const uint32_t take = cnt < 64u - off ? cnt : 64u - off;
            if (lane >= off && lane < off + take) { e = bg + (lane - off); tag = (uint32_t)k; tvalid = true; }
            tags |= 1ull << k;
            off += take; bg += take; cnt -= take;
            if (off == 64u) {
                H.guard--;
                heavy_tiny_batch<SHADOW, EXEC>(A, R, primary, tvalid, (size_t)e, tag, tags, T, H, lane);
                off = 0; tvalid = false; tags = 0;
            }
        }
    }
    if (off) heavy_tiny_batch<SHADOW, EXEC>(A, R, primary, tvalid, (size_t)e, tag, tags, T, H, lane);
}

// closest hit of the (wave-uniform) ray in one mesh
template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_mesh(const KernelArgs &A, const Ray &R, const bool primary, const uint32_t mesh,
                                           HeavyState &H, const uint32_t lane) {
    const HeavyMesh M = A.s->hmesh[mesh];
    const uint32_t guard = H.guard;
    H.guard = guard;
    H.mhave = false;
    H.mmin = INFINITY;
    H.mt = 0;
    H.mtri = 0;
    switch (M.n_levels) {
        case 1: heavy_chunk<0, SHADOW, EXEC>(A, R, primary, M, 0, heavy_chunk_load<0>(A, M, 0, lane), H, lane); break;
        case 2: heavy_chunk<1, SHADOW, EXEC>(A, R, primary, M, 0, heavy_chunk_load<1>(A, M, 0, lane), H, lane); break;
        case 3: heavy_chunk<2, SHADOW, EXEC>(A, R, primary, M, 0, heavy_chunk_load<2>(A, M, 0, lane), H, lane); break;
        case 4: heavy_chunk<3, SHADOW, EXEC>(A, R, primary, M, 0, heavy_chunk_load<3>(A, M, 0, lane), H, lane); break;
        default: break;
    }
}

// The top-level tree in registers.  Walked from memory it is ~40 dependent scalar loads per ray (nodes, leaf entries, mesh
// records) -- measured: half of a ray's 100 us in this kernel -- for a few hundred bytes that do not depend on the ray.  When
// crt_create finds it small enough (KernelArgs::top_fast) every wave loads it once: lane i keeps top-level node i, lane j
// leaf entry j, lane m the flags / pad of mesh m; a ray then tests all the nodes' boxes in one instruction and the walk
// reads links and entries with v_readlane.
struct TopRegs {
    float4 q0, q1;           // lane i: node top_first + i  {lo, miss} {hi, link}
    uint32_t entry, entry2;  // lane j: leaf_meshes[j], leaf_meshes[64 + j]
    uint32_t mflags, mpad;   // lane m: meshes[m].flags, meshes[m].pad
};
__device__ __forceinline__ TopRegs heavy_top_load(const KernelArgs &A, const uint32_t lane) {
    TopRegs TR;
    TR.q0 = make_float4(0, 0, 0, 0); TR.q1 = TR.q0; TR.entry = 0; TR.entry2 = 0; TR.mflags = 0; TR.mpad = 0;
    if (A.s->top_fast) {
        if (lane < A.s->top_count) { TR.q0 = A.s->nodes[2 * (size_t)(A.s->top_first + lane)]; TR.q1 = A.s->nodes[2 * (size_t)(A.s->top_first + lane) + 1]; }
        if (lane < A.s->top_leaf_entries) TR.entry = A.s->leaf_meshes[lane];
        if (64u + lane < A.s->top_leaf_entries) TR.entry2 = A.s->leaf_meshes[64u + lane];
        if (lane < A.s->top_meshes) { const crt_mesh m = A.s->meshes[lane]; TR.mflags = m.flags; TR.mpad = m.pad; }
    }
    return TR;
}

// The two-level walk for one ray per wave.  SHADOW: AccelerationStructure.cpp:56-94, else KDTree.cpp:127-167.
// The top-level tree comes in one of two forms: in registers (TopRegs, <= 64 nodes), or as its LEAF SEQUENCE (SceneArgs::plan_boxes,
// any size: the leaves in visit order, 64 boxes per instruction -- the argument at the top of this file holds for the top-level
// tree as it does for a mesh's).  (A tree with neither form -- its nodes not one index range -- keeps its rays in the per-lane kernels.)
template <bool SHADOW, bool EXEC>
__device__ __forceinline__ void heavy_walk(const KernelArgs &A, TopRegs &TR, const Ray &R, const bool primary, const float light_dist,
                                           bool &have, float &bt, uint32_t &btri, uint32_t &bmesh, bool &occluded,
                                           const uint32_t lane) {
    have = false;
    occluded = false;
    float tmin = INFINITY;
    SeenMeshes seen;
    seen_clear(seen);
    HeavyState H;
    H.guard = 1u << 18;
    H.stop = false;
    H.light_dist = light_dist;
    H.nbox = 0; H.ntri = 0; H.n_chunks = 0; H.n_batches = 0; H.n_meshes = 0;
    TinyResults T;
    heavy_tiny_meshes<SHADOW, EXEC>(A, R, primary, T, H, lane);
    const bool fast = A.s->top_fast != 0;
    // one mesh of a top-level leaf's list, in list order; false: the walk is over (a shadow ray found its occluder)
    auto visit_mesh = [&](const uint32_t mi, const uint32_t mflags, const uint32_t mpad) -> bool {
        if ((SHADOW && (mflags & 1u) && !A.f->use_gi) || mesh_walk_is_repeat(seen, mi)) return true;  // (AccelerationStructure.cpp:66-71: not in the GI mode)  // (kernel_common.h: every mesh once per ray)
        if (mpad) {  // a single-leaf mesh: its result has been waiting in lane pad - 1 since the start of the ray
            const int k = (int)mpad - 1;
            H.mhave = __builtin_amdgcn_readlane((int)T.have, k) != 0;
            H.mt = lane_value(T.mt, k);
            H.mtri = lane_value(T.mtri, k);
            H.mmin = lane_value(T.mmin, k);
        } else {
            if (EXEC) H.n_meshes++;
            heavy_mesh<SHADOW, EXEC>(A, R, primary, mi, H, lane);
        }
        if (SHADOW && H.stop) { occluded = true; return false; }
        if (H.mhave) {
            if (SHADOW) {
                const float px = R.ox + R.dx * H.mt, py = R.oy + R.dy * H.mt, pz = R.oz + R.dz * H.mt;
                if (len3(px - R.ox, py - R.oy, pz - R.oz) <= light_dist) occluded = true;
            } else {
                if (!have) { have = true; bt = H.mt; btri = H.mtri; bmesh = mi; }
                if (H.mt < tmin) { tmin = H.mt; bt = H.mt; btri = H.mtri; bmesh = mi; }
            }
        }
        return true;
    };
    // The meshes of the top-level leaves the ray reaches, one per call, in the reference's order (all state wave-uniform but the
    // registers of TR, which in the leaf-sequence form hold the current 64 leaves and the current 64 list entries).
    uint32_t e = NONE;                                    // both forms: next entry of the current leaf's list (NONE: not in a leaf)
    uint32_t chunk = 0, left = 0, bi = 0, bn = 0;         // leaf-sequence form: next chunk, entries left in the list, batch cursor / size
    unsigned long long leaves = 0;                        // ... leaves of the current chunk that pass and are still to visit
    // registers form: every top-level box against this ray in one instruction; the leaves among the nodes that pass are the leaves the
    // reference reaches (nested boxes: a leaf's box passes only if all its ancestors' do), in index order, which is visit order
    unsigned long long pending = fast ? __ballot(lane < A.s->top_count && is_leaf_link(__float_as_uint(TR.q1.w)) &&
                                                 slab_test(R, TR.q0.x, TR.q0.y, TR.q0.z, TR.q1.x, TR.q1.y, TR.q1.z)) : 0ull;
    if (EXEC && fast) H.nbox += A.s->top_count;
    auto next_mesh = [&](uint32_t &mi, uint32_t &mflags, uint32_t &mpad) -> bool {
        if (fast) {
            if (e == NONE) {
                if (!pending || !H.go()) return false;
                H.guard--;
                const int i = __ffsll((long long)pending) - 1;
                pending &= pending - 1ull;
                e = lane_value(__float_as_uint(TR.q1.w), i) & ~LEAF;
            }
            const uint32_t ent = e < 64u ? lane_value(TR.entry, (int)e) : lane_value(TR.entry2, (int)(e - 64u));
            e++;
            if (ent & LAST) e = NONE;
            mi = ent & ~LAST;
            mflags = lane_value(TR.mflags, (int)mi); mpad = lane_value(TR.mpad, (int)mi);
            return true;
        }
        for (;;) {
            if (!H.go()) return false;
            H.guard--;
            if (bi < bn) {
                mi = lane_value(TR.entry, (int)bi); mflags = lane_value(TR.mflags, (int)bi); mpad = lane_value(TR.mpad, (int)bi);
                bi++;
                return true;
            }
            if (left) {  // the next (up to) 64 entries of the list: lane j reads entry j and its mesh record
                bn = left < 64u ? left : 64u;
                TR.entry = lane < bn ? A.s->leaf_meshes[e + lane] & ~LAST : 0u;
                const crt_mesh mm = A.s->meshes[TR.entry];  // (lanes >= bn: mesh 0, not used)
                TR.mflags = mm.flags; TR.mpad = mm.pad;
                bi = 0; e += bn; left -= bn;
                continue;
            }
            if (leaves) {
                const int j = __ffsll((long long)leaves) - 1;
                leaves &= leaves - 1;
                e = lane_value(__float_as_uint(TR.q0.w), j); left = lane_value(__float_as_uint(TR.q1.w), j);
                continue;
            }
            if (chunk * 64u >= A.s->plan_leaves) return false;
            const uint32_t k = chunk * 64u + lane;  // lane j tests leaf 64 c + j: {lo, first entry} {hi, entries}
            const bool valid = k < A.s->plan_leaves;
            const float4 *L = reinterpret_cast<const float4 *>(A.s->plan_boxes) + 4 * (size_t)(valid ? k : 0u);
            TR.q0 = L[0]; TR.q1 = L[1];
            if (EXEC) H.nbox += (uint32_t)__popcll(__ballot(valid));
            leaves = __ballot(valid && slab_test(R, TR.q0.x, TR.q0.y, TR.q0.z, TR.q1.x, TR.q1.y, TR.q1.z));
            chunk++;
        }
    };
    for (;;) {
        uint32_t mi, mflags, mpad;
        if (!next_mesh(mi, mflags, mpad) || !visit_mesh(mi, mflags, mpad) || !H.go()) break;
    }
    if (!H.guard && lane == 0) { A.f->s_counts[SC_GUARD] = 1; A.f->s_counts[SC_OVERFLOW] = 1; }  // bound hit: let the fallback redo the frame
    if (EXEC && lane == 0) {
        atomicAdd(&A.exec_counters[0], (unsigned long long)H.nbox); atomicAdd(&A.exec_counters[1], (unsigned long long)H.ntri);
        uint32_t *diag = A.f->s_counts + SC_HEAVY_DIAG + (SHADOW ? 8 : 0);  // tools/stream_stats.py: what a wave-per-ray walk consists of
        atomicAdd(diag + 0, 1u); atomicAdd(diag + 1, H.n_chunks); atomicAdd(diag + 2, H.n_batches); atomicAdd(diag + 3, H.n_meshes);
        atomicAdd(diag + 4, (1u << 18) - H.guard);
    }
}

__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// closest hits of the rays evicted from stream_trace_shade(gen); results go to s_hits[k] for list entry k
template <bool GI, bool EXEC>
__device__ __forceinline__ void heavy_closest_level(const KernelArgs &A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);  // ahead of the bulk shadow pass's waves on this SIMD
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    const bool whole = stream_level_is_whole_heavy(A, gen, count);  // every ray of the level: entry k is ray k
    if (A.force_whole && !whole && gen > 0) {  // no per-lane launch and more rays than s_hits holds: the fallback redoes the frame
        if (threadIdx.x == 0 && blockIdx.x == 0) A.f->s_counts[SC_OVERFLOW] = 1;
        return;
    }
    uint32_t total = whole ? count : A.f->s_counts[SC_HEAVY + gen];
    if (total > A.f->s_heavy_cap) total = A.f->s_heavy_cap;
    const float4 *in_q = A.f->s_rayq[gen & 1u];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    if (wave >= total) return;
    TopRegs TR = heavy_top_load(A, lane);
    for (uint32_t k = wave; k < total; k += n_waves) {  // one evicted ray per wave and trip
        const uint32_t r = whole ? k : A.f->s_heavy[k];
        Ray R;
        bool primary = false;
        if (gen == 0) {
            level0_ray<GI>(A, level0_decode<GI>(A, r), R);
            primary = true;
        } else {
            const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
            R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
            R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
            normalize3(R.dx, R.dy, R.dz);
            ray_prepare(R);
        }
        // (the ray stays in vector registers, the same value in every lane: a vector instruction takes one scalar operand at most,
        //  and the scalar file is what this kernel runs out of)
        R.parmask = __builtin_amdgcn_readfirstlane(R.parmask);
        bool have = false, occluded = false;
        float bt = 0;
        uint32_t btri = 0, bmesh = 0;
        heavy_walk<false, EXEC>(A, TR, R, primary, 0.0f, have, bt, btri, bmesh, occluded, lane);
        if (lane == 0) A.f->s_hits[k] = make_float4(bt, __uint_as_float(btri), __uint_as_float(bmesh), __uint_as_float(have ? 1u : 0u));
    }
}
__global__ __launch_bounds__(BLOCK, 5) void heavy_trace_closest(const KernelArgs A, const uint32_t gen) { heavy_closest_level<false, false>(A, gen); }
__global__ __launch_bounds__(BLOCK, 5) void heavy_trace_closest_gi(const KernelArgs A, const uint32_t gen) { heavy_closest_level<true, false>(A, gen); }  // (level 0: jittered samples)
// the build that tallies the tests it executes (crt_options::collect_counters == 2): its five tallies are wave-uniform state the walk has no registers for
__global__ __launch_bounds__(BLOCK, 5) void heavy_trace_closest_tally(const KernelArgs A, const uint32_t gen) { heavy_closest_level<false, true>(A, gen); }

// part 0: the walks shadow pass 0 gave up (list entries below the SC_SHEAVY_SPLIT mark), on the side stream right
// after that pass; part 1: the rest, after the last pass.
template <bool EXEC>
__device__ __forceinline__ void heavy_shadow_part(const KernelArgs &A, const uint32_t part) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    uint32_t total = A.f->s_counts[SC_SHEAVY], split = A.f->s_counts[SC_SHEAVY_SPLIT];
    if (total > A.f->s_heavy_cap) total = A.f->s_heavy_cap;
    if (split > total) split = total;
    const uint32_t first = part == 0 ? 0u : split;
    if (part == 0) total = split;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    if (first + wave >= total) return;
    TopRegs TR = heavy_top_load(A, lane);
    for (uint32_t k = first + wave; k < total; k += n_waves) {
        const uint32_t r = A.f->s_sheavy[k];
        const float4 q0 = A.f->s_shadowq[2 * (size_t)r], q1 = A.f->s_shadowq[2 * (size_t)r + 1];
        Ray R;
        R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
        R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
        ray_prepare(R);
        R.parmask = __builtin_amdgcn_readfirstlane(R.parmask);
        bool have, occluded;
        float bt = 0;
        uint32_t btri = 0, bmesh = 0;
        heavy_walk<true, EXEC>(A, TR, R, false, uniform_f(q0.w), have, bt, btri, bmesh, occluded, lane);
        if (lane == 0) A.f->s_occluded[r] = occluded ? 1 : 0;
    }
}
__global__ __launch_bounds__(BLOCK) void heavy_trace_shadow(const KernelArgs A, const uint32_t part) { heavy_shadow_part<false>(A, part); }
__global__ __launch_bounds__(BLOCK) void heavy_trace_shadow_tally(const KernelArgs A, const uint32_t part) { heavy_shadow_part<true>(A, part); }
