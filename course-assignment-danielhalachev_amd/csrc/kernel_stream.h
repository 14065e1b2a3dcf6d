// kernel_stream.h -- the ray-stream form of the hot path (default).
//
// The reference evaluates a pixel as a depth-first recursion: shootRay -> calculateReflection /
// calculateRefraction -> shootRay ... with calculateDiffusion's shadow rays at the leaves
// (RayTracer.cpp:300-451).  Run that way on a GPU, ONE pixel's chain of up to 2^(MAX_DEPTH+1)-1 dependent
// tree walks is the critical path of the whole frame.  Here the recursion is unrolled by GENERATION instead:
//
//   stream_trace_shade(g), g = 0 .. MAX_DEPTH   every ray of recursion level g, one ray per lane: closest-hit
//                                               walk, then the material dispatch.  It records a node of the
//                                               pixel's ray tree (TNode), appends the child rays (level g+1)
//                                               to the next queue and the hit's shadow rays to the shadow queue.
//   stream_trace_shadow                         every shadow ray of every level in ONE launch -> occlusion flag.
//   stream_resolve                              per pixel, walks its ray tree in the reference's post-order and
//                                               combines the colours with the reference's own expressions, so
//                                               every float is produced by the same operations in the same order.
//
// All rays of a level are independent, so each launch has millions of rays of parallelism and no lane ever
// waits for a recursion.  Queues, the ray tree and the flags live in HBM (a few hundred MB at 1080p).
#pragma once

#include "kernel_common.h"
#include "kernel_lane.h"
#include "kernel_walk.h"

struct TNode {          // one node of a pixel's ray tree, 32 bytes
    float cx, cy, cz;   // DIFFUSE: texture/albedo colour; REFLECT: albedo; after resolve of a REFRACT node: reflection colour
    uint32_t kind;
    uint32_t a, b;      // DIFFUSE: first shadow-ray index, stride between lights; REFLECT: child; REFRACT: reflection child, refraction child
    float f;            // REFRACT: Fresnel coefficient
    uint32_t pad;
};
enum : uint32_t { TN_CONST = 0, TN_DIFFUSE = 1, TN_REFLECT = 2, TN_REFRACT = 3, TN_SKIP = 4, TN_KIND_MASK = 0xFFu,
                  TN_BITMAP = 0x100u, TN_PARKED = 0x200u,  // stream_resolve: a partial result is parked in the node
                  // the GI mode (RayTracer.cpp:331-354): a DIFFUSE node with gi_samples children at consecutive node indices
                  // from `f` on (CHILD_BG: beyond MAX_DEPTH, background each); resolve counts the finished ones in bits 16..23
                  TN_GI = 0x400u, TN_GI_DONE_SHIFT = 16 };
constexpr uint32_t CHILD_BG = 0xFFFFFFFFu;     // child ray beyond MAX_DEPTH: background without tracing (RayTracer.cpp:427-429)
constexpr uint32_t CHILD_NONE = 0xFFFFFFFEu;   // total internal reflection: no refraction child (RayTracer.cpp:416)
// (MAX_GENERATIONS, the SC_* layout of the counter block and the level queue's LQ_* words: kernel_common.h -- the host side reads them too)

// Level 0 holds level0_samples rays per pixel (the GI mode's RAYS_PER_PIXEL, RayTracer.cpp:90-104; otherwise one): ray r is
// pixel r & 63 of work item (r >> 6) % n_items, sample (r >> 6) / n_items -- the samples are whole copies of the item list, so
// that 64 consecutive rays are still one 8x8 tile and the fixed shadow slots (below) keep their layout.
__device__ __forceinline__ uint32_t stream_level_count(const KernelArgs &A, uint32_t g) {
    return g == 0 ? A.f->n_items * 64u * A.f->level0_samples : A.f->s_counts[SC_COUNT + g];
}
struct Level0Ray { uint32_t item, sub, sample, px, py; bool covered; };
template <bool GI>  // (the kernels of the GI mode are builds of their own: the plain ones carry none of its code)
__device__ __forceinline__ Level0Ray level0_decode(const KernelArgs &A, const uint32_t r) {
    Level0Ray P;
    const uint32_t slot = r >> 6;
    P.sample = GI && A.f->level0_samples > 1u ? slot / A.f->n_items : 0u;
    P.item = slot - P.sample * A.f->n_items;
    P.sub = r & 63u;
    const WorkItem wi = A.f->items[P.item];
    P.px = (wi.tile % A.s->tiles_x) * TILE + (P.sub & 7u);
    P.py = (wi.tile / A.s->tiles_x) * TILE + (P.sub >> 3);
    P.covered = ((wi.mask >> P.sub) & 1ull) && P.px < A.s->width && P.py < A.s->height;
    return P;
}
// the key of the shootRay invocation of level-0 ray P (gi_random.h)
__device__ __forceinline__ uint32_t level0_key(const KernelArgs &A, const Level0Ray &P) {
    return crt_gi_mix(crt_gi_mix(A.f->gi_seed, P.py * A.s->width + P.px), P.sample);
}
// getRay for it: the pixel centre, or -- samples 1 .. RAYS_PER_PIXEL-1 -- a jittered position (RayTracer.cpp:61-80, 96)
template <bool GI>
__device__ __forceinline__ void level0_ray(const KernelArgs &A, const Level0Ray &P, Ray &R) {
    if (!GI || P.sample == 0u) { primary_ray(A, P.px, P.py, R); return; }
    if constexpr (GI) {
        const uint32_t key = level0_key(A, P);
        primary_ray_offset(A, P.px, P.py, crt_gi_uniform(key, 0u), crt_gi_uniform(key, 1u), R);
    }
}
// Levels below the threshold skip the per-lane kernel: list entry k of the wave-per-ray kernel is ray k itself.
// (force_whole: the host did not launch the per-lane kernel for this level at all -- a frame ago the level was far below the
// threshold, crt_tuning::level_grid -- so the wave-per-ray kernel takes every ray whatever the count turns out to be.)
__device__ __forceinline__ bool stream_level_is_whole_heavy(const KernelArgs &A, uint32_t g, uint32_t count) {
    return g > 0 && (A.force_whole || count < A.f->heavy_level_threshold) && count <= A.f->s_heavy_cap;  // s_hits holds s_heavy_cap records
}
__device__ __forceinline__ uint32_t stream_level_base(const KernelArgs &A, uint32_t g) {
    uint32_t base = 0;
    for (uint32_t k = 0; k < g; k++) base += stream_level_count(A, k);
    return base;
}

// wave-aggregated fetch of one work index per requesting lane (all lanes executing this must want one)
__device__ __forceinline__ uint32_t wave_fetch(uint32_t *cursor, uint32_t lane) {
    const unsigned long long need = __ballot(1);
    const int rank = __popcll(need & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (rank == 0) base = atomicAdd(cursor, (uint32_t)__popcll(need));
    base = __shfl(base, __ffsll((long long)need) - 1);
    return base + (uint32_t)rank;
}

// The same for the two launches with MILLIONS of indices to hand out (level 0's primary rays, the bulk shadow pass's slots): a wave
// claims a CHUNK of consecutive indices with ONE atomic and serves its next refills from it.  One cursor word serves ~100-150 atomics
// per microsecond whatever the number of waves asking: 8.3 M shadow slots claimed ~28 at a time were 300 k atomics -- the pass took
// 1.9 ms on one, two or five workgroups per CU alike, 1.4 claimed 256 at a time.  Chunks shrink towards the end of the launch (a wave
// takes at most its share of HALF of what it sees left, never less than 64): beside the recursion levels the pass's waves advance at very
// different rates, and whole chunks of 512 left it 0.5 ms of stragglers.  (Indices past the end are claimed and dropped.)
struct WaveChunk { uint32_t next, end; };   // wave-uniform
// (EVERY lane of the wave calls this, `want` or not: the chunk is the wave's, and a lane that skipped the call would keep a stale copy)
__device__ __forceinline__ uint32_t wave_fetch_chunked(uint32_t *cursor, uint32_t lane, bool want, WaveChunk &C, const uint32_t chunk_max, const uint32_t total) {
    const unsigned long long need = __ballot(want);
    const uint32_t n = (uint32_t)__popcll(need);
    const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
    const uint32_t avail = C.end - C.next;
    if (avail >= n) {   // (wave-uniform)
        const uint32_t idx = C.next + rank;
        C.next += n;
        return idx;
    }
    // what is left, as of this wave's own previous claim (C.end: reading the cursor word itself would be one more request to the one
    // channel the claims queue up at -- measured: the frame 3.5 -> 5.9 ms)
    const uint32_t left = C.end < total ? total - C.end : 0u;
    const uint32_t share = (left / (2u * gridDim.x * (BLOCK / 64u))) & ~63u;
    const uint32_t c = share > chunk_max ? chunk_max : (share < 64u ? 64u : share);   // (>= 64: a refill of all 64 lanes fits in one claim)
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(cursor, c);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t idx = rank < avail ? C.next + rank : base + (rank - avail);   // (the old chunk's last indices first)
    C.next = base + (n - avail);
    C.end = base + c;
    return idx;
}

// wave-aggregated append of this lane's ray id to an eviction list; false when the list is full
__device__ __forceinline__ bool evict_ray(uint32_t *list, uint32_t cap, uint32_t *count, uint32_t r, uint32_t lane) {
    const unsigned long long em = __ballot(1);
    uint32_t base = 0;
    if ((em & ((1ull << lane) - 1ull)) == 0) base = atomicAdd(count, (uint32_t)__popcll(em));
    base = __shfl(base, __ffsll((long long)em) - 1);
    const uint32_t slot = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
    if (slot >= cap) return false;
    list[slot] = r;
    return true;
}

// Level 0 does not queue its shadow rays, it has fixed slots for them, and no atomic is needed: 64 consecutive slots are one 8x8
// tile's rays towards one light -- rays that visit the same nodes at the same time, which is what the vector L1 serves cheaply --
// and the tiles come in blocks of LEVEL0_SHADOW_BLOCK consecutive work items, LIGHT-MAJOR inside a block: {16 tiles towards light 0}
// {the same 16 towards light 1} ...  Waves take consecutive 64-slot batches at about the same time, so the waves that are in flight
// together walk in one direction from neighbouring places (measured against tile-major, one tile's lights in turn: HW14 354 -> 367,
// HW12 436 -> 473 Mpixels/s; blocks of 4 .. 240 tiles alike, the whole frame light by light 349 / 452).  Pixel r of light li owns slot
// base + li * stride + (r & 63); the last block of a frame is as long as the items left.  A pixel without a diffuse hit marks its
// slots unused (distance word all ones: no computed distance has that pattern).
constexpr uint32_t SHADOW_SLOT_UNUSED = 0xFFFFFFFFu;
constexpr uint32_t LEVEL0_SHADOW_BLOCK = 16;
__device__ __forceinline__ void level0_shadow_place(const KernelArgs &A, uint32_t r, uint32_t &base, uint32_t &stride) {
    const uint32_t item = r >> 6, total = A.f->n_items * A.f->level0_samples;
    const uint32_t blk = item / LEVEL0_SHADOW_BLOCK, in = item - blk * LEVEL0_SHADOW_BLOCK;
    const uint32_t left = total - blk * LEVEL0_SHADOW_BLOCK, nb = left < LEVEL0_SHADOW_BLOCK ? left : LEVEL0_SHADOW_BLOCK;
    base = blk * LEVEL0_SHADOW_BLOCK * A.s->n_lights * 64u + in * 64u;
    stride = nb * 64u;
}
__device__ __forceinline__ void level0_release_shadow_slots(const KernelArgs &A, uint32_t r) {
    if (!A.f->fixed0) return;
    uint32_t base, stride;
    level0_shadow_place(A, r, base, stride);
    const size_t first = (size_t)base + (r & 63u);
    for (uint32_t li = 0; li < A.s->n_lights; li++)
        A.f->s_shadowq[2 * (first + (size_t)li * stride)] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(SHADOW_SLOT_UNUSED));
}

// shootRay's dispatch on the closest hit (RayTracer.cpp:431-450), in two steps.  shade_hit: everything but the child
// rays' queue slots -- the ray-tree node's contents, the shadow rays (appended to the shadow queue), and the child rays
// themselves (reflection, and transmission for a refractive hit without total internal reflection).  The caller then
// allocates the children and stores the node (shade_and_emit).
struct Shaded {
    TNode N;               // N.a / N.b: CHILD_BG / CHILD_NONE / 0 until the caller fills in the children's node indices
    bool reflect, transmit;  // child rays to trace (false beyond MAX_DEPTH: such a child is background without tracing)
    float rox, roy, roz, rdx, rdy, rdz;   // reflection ray (both mirror materials)
    float tox, toy, toz, tdx, tdy, tdz;   // transmission ray
    bool gi;                               // a diffuse hit in the GI mode whose gi_samples child rays are to be traced
    float hpx, hpy, hpz, hnx, hny, hnz;   // ... its hit point and normal (the sample directions: gi_sample_direction)
};

// Called by every lane whose walk has just ended (any subset of the wave); shadow-slot allocation is aggregated over them.
template <bool COUNT, bool GI>
__device__ __forceinline__ void shade_hit(const KernelArgs &A, const uint32_t gen, const uint32_t r, const Ray &R, const bool have,
                                          const float bt, const uint32_t btri, const uint32_t bmesh, uint32_t *cnt,
                                          const uint32_t lane, Shaded &E, bool *out_diffuse = nullptr,
                                          uint32_t *out_first = nullptr, uint32_t *out_stride = nullptr) {
    if (out_diffuse) *out_diffuse = false;
    const bool spawn_allowed = gen + 1 <= A.f->max_depth;  // a child enters shootRay with depth gen+1 (RayTracer.cpp:427)
    TNode &N = E.N;
    N.cx = A.s->bgx; N.cy = A.s->bgy; N.cz = A.s->bgz;
    N.kind = TN_CONST; N.a = 0; N.b = 0; N.f = 0; N.pad = 0;
    E.reflect = false; E.transmit = false; E.gi = false;
    E.hpx = E.hpy = E.hpz = E.hnx = E.hny = E.hnz = 0;
    E.rox = E.roy = E.roz = E.rdx = E.rdy = E.rdz = 0;
    E.tox = E.toy = E.toz = E.tdx = E.tdy = E.tdz = 0;
    if (have) {
        Surface S;
        surface_at(A, R, bt, btri, bmesh, S);
        if (COUNT) cnt[C_HIT]++;
        if (S.M.type == CRT_MAT_DIFFUSE) {
            // calculateDiffusion (RayTracer.cpp:300-330): the light loop becomes n_lights shadow rays
            bool bitmap = false;
            if (S.M.texture >= 0)
                texture_color<COUNT>(A, A.s->textures[S.M.texture], btri, S.u, S.v, 1.0f - S.u - S.v, N.cx, N.cy, N.cz, bitmap);
            else { N.cx = S.M.ax; N.cy = S.M.ay; N.cz = S.M.az; }
            const unsigned long long mask = __ballot(1);
            uint32_t cntd = (uint32_t)__popcll(mask);
            uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            uint32_t base = 0;
            if (gen == 0 && A.f->fixed0) {
                level0_shadow_place(A, r, base, cntd);
                rank = r & 63u;
            } else {
                if (rank == 0) base = atomicAdd(A.f->s_counts + SC_SHADOW, cntd * A.s->n_lights);
                base = __shfl(base, __ffsll((long long)mask) - 1);
            }
            if (!(gen == 0 && A.f->fixed0) && (uint64_t)base + (uint64_t)cntd * A.s->n_lights > A.f->s_shadow_cap) {
                A.f->s_counts[SC_OVERFLOW] = 1;
                N.cx = N.cy = N.cz = 0;  // the frame is redone by the fallback path
            } else {
                N.kind = TN_DIFFUSE | (bitmap ? TN_BITMAP : 0u) | (GI ? TN_GI : 0u);
                N.a = base + rank;
                N.b = cntd;
                if constexpr (GI) {  // the indirect term's child rays enter shootRay at depth gen + 1 (RayTracer.cpp:349-350)
                    N.f = __uint_as_float(CHILD_BG);
                    E.gi = spawn_allowed && A.f->gi_samples > 0u;
                    E.hpx = S.px; E.hpy = S.py; E.hpz = S.pz; E.hnx = S.nx; E.hny = S.ny; E.hnz = S.nz;
                }
                if (out_diffuse) { *out_diffuse = true; *out_first = N.a; *out_stride = cntd; }
                for (uint32_t li = 0; li < A.s->n_lights; li++) {
                    Ray SR;
                    float dist, kfac;
                    light_setup(A, li, S.px, S.py, S.pz, S.nx, S.ny, S.nz, SR, dist, kfac);
                    if (COUNT) { cnt[C_LIGHT]++; cnt[C_SHADOW]++; }
                    const size_t slot = (size_t)base + (size_t)li * cntd + rank;
                    A.f->s_shadowq[2 * slot] = make_float4(SR.ox, SR.oy, SR.oz, dist);
                    A.f->s_shadowq[2 * slot + 1] = make_float4(SR.dx, SR.dy, SR.dz, kfac);
                }
            }
        } else if (S.M.type == CRT_MAT_REFLECTIVE || S.M.type == CRT_MAT_REFRACTIVE) {
            const bool refractive = S.M.type == CRT_MAT_REFRACTIVE;
            float nx = S.nx, ny = S.ny, nz = S.nz;
            bool transmit = false;
            if (refractive) {
                // calculateRefraction (RayTracer.cpp:375-417)
                float eta1 = 1.0f, eta2 = S.M.ior;
                float idn = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
                if (idn > 0) {
                    const float s = eta1; eta1 = eta2; eta2 = s;
                    nx = -1.0f * nx; ny = -1.0f * ny; nz = -1.0f * nz;
                    idn = -idn;
                }
                const float cos_a = -idn;
                const float sin_a = sqrtf(std_max(0.0f, 1 - cos_a * cos_a));
                const float eta_ratio = eta1 / eta2;
                const float sin_b = eta_ratio * sin_a;
                if (sin_b < 1.0f) {
                    const float q = (eta1 - eta2) / (eta1 + eta2);
                    const float r0 = q * q;  // std::powf(q, 2), folded to q*q by the reference's compiler at -O2
                    N.f = r0 + (1 - r0) * crt_pow5(1.0f - cos_a);
                    const float cos_b = sqrtf(std_max(0.0f, 1 - sin_b * sin_b));
                    E.tdx = eta_ratio * (R.dx + cos_a * nx) - cos_b * nx;
                    E.tdy = eta_ratio * (R.dy + cos_a * ny) - cos_b * ny;
                    E.tdz = eta_ratio * (R.dz + cos_a * nz) - cos_b * nz;
                    normalize3(E.tdx, E.tdy, E.tdz);
                    E.tox = S.px - nx * A.f->refraction_bias; E.toy = S.py - ny * A.f->refraction_bias; E.toz = S.pz - nz * A.f->refraction_bias;
                    transmit = true;
                }
                N.kind = TN_REFRACT;
            } else {
                N.kind = TN_REFLECT;  // calculateReflection (RayTracer.cpp:358-374)
                N.cx = S.M.ax; N.cy = S.M.ay; N.cz = S.M.az;
            }
            // the reflection ray (both materials): origin + n*bias, reflect(d, n) normalised (Vector.cpp:119-122)
            const float k = 2 * dot3(R.dx, R.dy, R.dz, nx, ny, nz);
            E.rdx = R.dx - k * nx; E.rdy = R.dy - k * ny; E.rdz = R.dz - k * nz;
            normalize3(E.rdx, E.rdy, E.rdz);
            E.rox = S.px + nx * A.f->reflection_bias; E.roy = S.py + ny * A.f->reflection_bias; E.roz = S.pz + nz * A.f->reflection_bias;
            N.a = CHILD_BG;
            N.b = refractive ? (transmit ? CHILD_BG : CHILD_NONE) : 0u;
            E.reflect = spawn_allowed;
            E.transmit = spawn_allowed && transmit;
        }
        // any other material type (Constant): background, RayTracer.cpp:443-446
    }
    if (gen == 0 && (N.kind & TN_KIND_MASK) != TN_DIFFUSE) level0_release_shadow_slots(A, r);
}

__device__ __forceinline__ void store_tnode(const KernelArgs &A, const size_t index, const TNode &N) {
    float4 *dst = A.f->s_nodes + 2 * index;
    dst[0] = make_float4(N.cx, N.cy, N.cz, __uint_as_float(N.kind));
    dst[1] = make_float4(__uint_as_float(N.a), __uint_as_float(N.b), N.f, 0.0f);
}

// A child ray in a closest-hit queue: {origin, key of its shootRay invocation (GI mode; gi_random.h)} {direction, index of its ray-tree node}
__device__ __forceinline__ void store_child_ray(float4 *q, const size_t index, const float ox, const float oy, const float oz,
                                                const float dx, const float dy, const float dz, const uint32_t key,
                                                const uint32_t node) {
    q[2 * index] = make_float4(ox, oy, oz, __uint_as_float(key));
    q[2 * index + 1] = make_float4(dx, dy, dz, __uint_as_float(node));
}

// The LEVEL QUEUE (kernel_bvh.h: bvh_trace_queue): every ray below level 0 in ONE queue, whatever its level; a ray is consumed by
// another workgroup -- possibly on another XCD, whose L2 is not coherent with the producer's -- while the launch that produced it
// still runs.  A record is therefore eight 8-byte granules {tag = the frame's epoch, value}, each ONE agent-scope (write-through)
// store: the data is its own flag, a consumer re-reads a record with agent-scope loads until every tag is the frame's epoch
// (/opt/skills/guides/cdna_hip_programming.md, Guideline 16, R2).  Values: origin x y z, level, direction x y z, spare.
typedef unsigned long long __attribute__((address_space(1))) *lq_ptr;
__device__ __forceinline__ void lq_store_ray(const KernelArgs &A, const uint32_t index, const float ox, const float oy, const float oz,
                                             const float dx, const float dy, const float dz, const uint32_t level) {
    lq_ptr g = (lq_ptr)(A.f->s_lq + 8 * (size_t)index);
    const unsigned long long tag = (unsigned long long)A.f->lq_epoch << 32;
    const uint32_t v[8] = {__float_as_uint(ox), __float_as_uint(oy), __float_as_uint(oz), level, __float_as_uint(dx), __float_as_uint(dy), __float_as_uint(dz), 0u};
#pragma unroll
    for (int k = 0; k < 8; k++) __hip_atomic_store(g + k, tag | v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// An entry whose ray its parent's lane walks itself (bvh_trace_queue: the reflection child continues in the lane that shaded the hit): the
// entry exists -- it owns the ray's node and is counted -- but whoever claims it has nothing to do: granule 7 carries the mark.
__device__ __forceinline__ void lq_store_taken(const KernelArgs &A, const uint32_t index) {
    lq_ptr g = (lq_ptr)(A.f->s_lq + 8 * (size_t)index);
    __hip_atomic_store(g + 7, ((unsigned long long)A.f->lq_epoch << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
enum : int { LQ_NOT_YET = 0, LQ_RAY = 1, LQ_TAKEN = 2 };
// (one granule is looked at first: a lane that waits for its record asks for eight bytes per look, not sixty-four)
__device__ __forceinline__ int lq_load_ray(const KernelArgs &A, const uint32_t index, float &ox, float &oy, float &oz, float &dx, float &dy,
                                           float &dz, uint32_t &level) {
    lq_ptr g = (lq_ptr)(A.f->s_lq + 8 * (size_t)index);
    const unsigned long long g7 = __hip_atomic_load(g + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(g7 >> 32) != A.f->lq_epoch) return LQ_NOT_YET;
    if ((uint32_t)g7 == 1u) return LQ_TAKEN;
    unsigned long long x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) ok = ok && (uint32_t)(x[k] >> 32) == A.f->lq_epoch;
    ox = __uint_as_float((uint32_t)x[0]); oy = __uint_as_float((uint32_t)x[1]); oz = __uint_as_float((uint32_t)x[2]); level = (uint32_t)x[3];
    dx = __uint_as_float((uint32_t)x[4]); dy = __uint_as_float((uint32_t)x[5]); dz = __uint_as_float((uint32_t)x[6]);
    return ok ? LQ_RAY : LQ_NOT_YET;
}

// shade_hit + the per-level queues: ray `r` of level `gen` writes node node_base + r, its children are appended to the
// queue of level gen+1 (child k of that queue owns node child_base + k).  Allocations are aggregated over the calling lanes:
// the reflection rays, then the transmission rays, then -- GI mode -- gi_samples consecutive rays per diffuse hit.
// LQ: the children go to the level queue instead (entry k of it owns node child_base + k, whatever its level).
template <bool COUNT, bool GI, bool LQ = false>
__device__ __forceinline__ void shade_and_emit(const KernelArgs &A, const uint32_t gen, const uint32_t r, const uint32_t node_base,
                                       const uint32_t child_base, const Ray &R, const bool have, const float bt,
                                       const uint32_t btri, const uint32_t bmesh, uint32_t *cnt, const uint32_t lane,
                                       bool *out_diffuse = nullptr, uint32_t *out_first = nullptr, uint32_t *out_stride = nullptr,
                                       float *keep = nullptr, uint32_t *keep_index = nullptr) {
    // keep (level queue only): the reflection child is not written to the queue -- its entry is marked taken -- but handed back to the
    // caller, whose lane walks it next: {origin, direction} in keep[0..5], its entry in *keep_index (NONE: there is no such child)
    if (keep_index) *keep_index = NONE;
    Shaded E;
    shade_hit<COUNT, GI>(A, gen, r, R, have, bt, btri, bmesh, cnt, lane, E, out_diffuse, out_first, out_stride);
    constexpr bool gi_mode = GI;
    if (E.reflect || (GI && E.gi)) {
        float4 *out_q = A.f->s_rayq[(gen + 1u) & 1u];
        uint32_t *out_count = LQ ? A.f->s_lq_words + LQ_TAIL : A.f->s_counts + SC_COUNT + gen + 1;
        const uint32_t out_cap = LQ ? A.f->s_lq_cap : A.f->s_ray_cap;
        // this invocation's key: the children's keys and the sample directions' random numbers derive from it
        uint32_t key = 0;
        if constexpr (GI) key = gen == 0 ? level0_key(A, level0_decode<true>(A, r)) : __float_as_uint(A.f->s_rayq[gen & 1u][2 * (size_t)r].w);
        // wave-aggregated append of 1 or 2 (mirror materials) or gi_samples (diffuse, GI mode) child rays per lane
        const unsigned long long m0 = __ballot(1), m1 = __ballot(E.reflect), m2 = __ballot(E.transmit), m3 = GI ? __ballot(E.gi) : 0ull;
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2), n3 = GI ? (uint32_t)__popcll(m3) * A.f->gi_samples : 0u;
        uint32_t base = 0;
        if ((m0 & below) == 0) base = atomicAdd(out_count, n1 + n2 + n3);
        base = __shfl(base, __ffsll((long long)m0) - 1);
        if constexpr (LQ) {
            // the levels' sizes (what the queues are sized by, what the tools print): the calling lanes may be of different levels --
            // one addition per level present among them
            unsigned long long left = m0;
            while (left) {
                const uint32_t g = (uint32_t)__shfl((int)gen, __ffsll((long long)left) - 1);
                const unsigned long long same = __ballot(gen == g) & left;
                const uint32_t kids = (uint32_t)__popcll(m1 & same) + (uint32_t)__popcll(m2 & same);
                if (kids && lane == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(A.f->s_counts + SC_COUNT + g + 1, kids);
                left &= ~same;
            }
        }
        if ((uint64_t)base + n1 + n2 + n3 > out_cap || (uint64_t)child_base + base + n1 + n2 + n3 > A.f->s_node_cap) {
            // (level queue: entries were reserved that will never be written; the waves waiting for them watch this word, from other XCDs too)
            if (LQ) __hip_atomic_store((uint32_t __attribute__((address_space(1))) *)(A.f->s_lq_words + LQ_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.f->s_counts[SC_OVERFLOW] = 1;
        } else if (E.reflect) {
            const uint32_t i1 = base + (uint32_t)__popcll(m1 & below);
            if constexpr (LQ) {
                if (keep) {
                    lq_store_taken(A, i1);
                    keep[0] = E.rox; keep[1] = E.roy; keep[2] = E.roz; keep[3] = E.rdx; keep[4] = E.rdy; keep[5] = E.rdz;
                    *keep_index = i1;
                } else lq_store_ray(A, i1, E.rox, E.roy, E.roz, E.rdx, E.rdy, E.rdz, gen + 1u);
            }
            else store_child_ray(out_q, i1, E.rox, E.roy, E.roz, E.rdx, E.rdy, E.rdz, gi_mode ? crt_gi_child_key(key, 0u) : gen + 1u, child_base + i1);
            E.N.a = child_base + i1;
            if (E.transmit) {
                const uint32_t i2 = base + n1 + (uint32_t)__popcll(m2 & below);
                if constexpr (LQ) lq_store_ray(A, i2, E.tox, E.toy, E.toz, E.tdx, E.tdy, E.tdz, gen + 1u);
                else store_child_ray(out_q, i2, E.tox, E.toy, E.toz, E.tdx, E.tdy, E.tdz, gi_mode ? crt_gi_child_key(key, 1u) : gen + 1u, child_base + i2);
                E.N.b = child_base + i2;
            }
        } else if constexpr (GI) {
            // RayTracer.cpp:333-350: sample i's direction from the invocation's numbers 2 + 2i, 3 + 2i, its origin off the surface
            const uint32_t i3 = base + n1 + n2 + (uint32_t)__popcll(m3 & below) * A.f->gi_samples;
            for (uint32_t i = 0; i < A.f->gi_samples; i++) {
                float dx, dy, dz;
                gi_sample_direction(R.dx, R.dy, R.dz, E.hnx, E.hny, E.hnz, crt_gi_uniform(key, 2u + 2u * i), crt_gi_uniform(key, 3u + 2u * i), dx, dy, dz);
                store_child_ray(out_q, i3 + i, E.hpx + E.hnx * A.f->monte_carlo_bias, E.hpy + E.hny * A.f->monte_carlo_bias,
                                E.hpz + E.hnz * A.f->monte_carlo_bias, dx, dy, dz, crt_gi_child_key(key, 2u + i), child_base + i3 + i);
            }
            E.N.f = __uint_as_float(child_base + i3);
        }
    }
    store_tnode(A, (size_t)node_base + r, E.N);
}

template <bool COUNT, bool GI = false>
__global__ __launch_bounds__(BLOCK) void stream_trace_shade(const KernelArgs A, const uint32_t gen) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;  // the fallback path redoes the frame
    const uint32_t count = stream_level_count(A, gen);
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const float4 *in_q = A.f->s_rayq[gen & 1u];

    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;

    Ray R;
    LaneWalk L;
    int state = ST_FETCH;
    uint32_t r = 0;
    uint32_t steps = 0;

    for (;;) {
        // ------------------------------------------------------------------ fetch
        if (__ballot(state == ST_FETCH)) {
            while (state == ST_FETCH) {
                r = wave_fetch(A.f->s_counts + SC_FETCH + gen, lane);
                if (r >= count) { state = ST_DONE; break; }
                if (gen == 0) {
                    const Level0Ray P = level0_decode<GI>(A, r);
                    if (!P.covered) {
                        reinterpret_cast<uint32_t *>(A.f->s_nodes + 2 * (size_t)r)[3] = TN_SKIP;
                        level0_release_shadow_slots(A, r);
                        continue;
                    }
                    level0_ray<GI>(A, P, R);
                    L.rtype = RAY_PRIMARY;
                    if (COUNT) cnt[C_PRIMARY]++;
                } else {
                    const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
                    R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                    R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
                    normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                    ray_prepare(R);
                    L.rtype = RAY_REFLECTION;      // reflection and refraction rays behave alike in the walk
                    if (COUNT) cnt[C_SECONDARY]++;
                }
                traversal_begin(L, A.s->top_root);
                state = ST_TRAVERSE;
            }
        }
        if (!__ballot(state != ST_DONE)) break;

        // ------------------------------------------------------------------ traverse
        bool finished = false;
        if (state == ST_TRAVERSE) {
            int budget = 64;
            do {
                if (!traversal_step<COUNT>(L, R, A, cnt)) { finished = true; break; }
                steps++;
            } while (--budget > 0);
        }

        // ------------------------------------------------------------------ shade, or hand a long walk to heavy_trace
        if (finished) {
            shade_and_emit<COUNT, GI>(A, gen, r, node_base, child_base, R, L.have, L.bt, L.btri, L.bmesh, cnt, lane);
            steps = 0;
            state = ST_FETCH;
        } else if (state == ST_TRAVERSE && A.step_budget && steps >= A.step_budget) {
            const unsigned long long em = __ballot(1);
            uint32_t base = 0;
            if ((em & ((1ull << lane) - 1ull)) == 0) base = atomicAdd(A.f->s_counts + SC_HEAVY + gen, (uint32_t)__popcll(em));
            base = __shfl(base, __ffsll((long long)em) - 1);
            const uint32_t slot = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
            if (slot < A.f->s_heavy_cap) {
                A.f->s_heavy[slot] = r;
                state = ST_FETCH;
            }
            steps = 0;  // list full: keep walking here
        }
    }

    if (COUNT) {
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters[k], v);
        }
    }
}

// Shading of the rays stream_trace_shade(gen) evicted, after heavy_trace_closest(gen) has found their hits.
template <bool COUNT, bool GI = false>
__global__ __launch_bounds__(BLOCK) void stream_shade_evicted(const KernelArgs A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);  // ahead of the bulk shadow pass's waves on this SIMD
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    const bool whole = stream_level_is_whole_heavy(A, gen, count);
    uint32_t total = whole ? count : A.f->s_counts[SC_HEAVY + gen];
    if (total > A.f->s_heavy_cap) total = A.f->s_heavy_cap;
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const float4 *in_q = A.f->s_rayq[gen & 1u];
    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < total; k += gridDim.x * BLOCK) {
        const uint32_t r = whole ? k : A.f->s_heavy[k];
        Ray R;
        if (gen == 0) {
            level0_ray<GI>(A, level0_decode<GI>(A, r), R);
        } else {
            const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
            R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
            R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
            normalize3(R.dx, R.dy, R.dz);
            ray_prepare(R);
        }
        const float4 h = A.f->s_hits[k];
        shade_and_emit<COUNT, GI>(A, gen, r, node_base, child_base, R, __float_as_uint(h.w) != 0, h.x, __float_as_uint(h.y),
                              __float_as_uint(h.z), cnt, lane);
    }
    if (COUNT) {
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters[k], v);
        }
    }
}

// Shading of every ray of level `gen` whose walk a walk-only kernel (kernel_plan.h: SPLIT) finished: one ray per lane, all lanes busy.
template <bool GI>
__global__ __launch_bounds__(BLOCK) void stream_shade_all(const KernelArgs A, const uint32_t gen) {
    if (A.wave_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;
    const uint32_t count = stream_level_count(A, gen);
    if (stream_level_is_whole_heavy(A, gen, count)) return;  // (no walk-only launch did anything: stream_shade_evicted has the level)
    const uint32_t node_base = stream_level_base(A, gen);
    const uint32_t child_base = node_base + count;
    const float4 *in_q = A.f->s_rayq[gen & 1u];
    for (uint32_t r = blockIdx.x * BLOCK + threadIdx.x; r < count; r += gridDim.x * BLOCK) {
        const float4 h = A.f->s_hits_all[r];
        if (__float_as_uint(h.w) >= 2u) continue;  // evicted (stream_shade_evicted shades it) or not a ray
        Ray R;
        if (gen == 0) level0_ray<GI>(A, level0_decode<GI>(A, r), R);
        else {
            const float4 q0 = in_q[2 * (size_t)r], q1 = in_q[2 * (size_t)r + 1];
            R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
            R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;
            normalize3(R.dx, R.dy, R.dz);
            ray_prepare(R);
        }
        shade_and_emit<false, GI>(A, gen, r, node_base, child_base, R, __float_as_uint(h.w) != 0, h.x, __float_as_uint(h.y), __float_as_uint(h.z), nullptr, lane);
    }
}

// What a frame zeroes before its first launch -- the counter block, the level queue's five words (each with three neighbours), the tallies --
// and the one word it presets (level 0's fixed shadow slots), in ONE launch: as eight memsets they were eight tiny launches of ~7 us
// each on the frame's stream, 60 us of a 2.6 ms frame.
__global__ void stream_frame_reset(uint32_t *counts, uint32_t *lq_words, unsigned long long *exec, const uint32_t shadow_preset) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)SC_ALLOC_WORDS; i += blockDim.x) counts[i] = (i == (uint32_t)SC_SHADOW) ? shadow_preset : 0u;
    if (lq_words && threadIdx.x < 20u) {
        const int line[5] = {LQ_TAIL, LQ_HEAD, LQ_DONE, LQ_ABORT, LQ_LEVEL0};
        lq_words[line[threadIdx.x >> 2] + (threadIdx.x & 3u)] = 0u;
    }
    if (exec && threadIdx.x < 6u) exec[threadIdx.x] = 0ull;
}

// Every shadow ray of the frame: RayTracer::hasIntersection in tree mode (RayTracer.cpp:507-517 ->
// AccelerationStructure.cpp:56-94).  Writes 1 to s_occluded[i] when the light is blocked.
// The shadow queue is traced in two passes so that the first can overlap the deeper recursion levels:
// pass 0 = the rays queued by level 0 (indices below the split mark), on a side stream as soon as level 0 is
// done; pass 1 = the rest, after the last level.
__global__ void stream_mark_split(const KernelArgs A, const uint32_t dst_word, const uint32_t src_word) {
    if (threadIdx.x == 0 && blockIdx.x == 0) A.f->s_counts[dst_word] = A.f->s_counts[src_word];
}

template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void stream_trace_shadow(const KernelArgs A, const uint32_t pass) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;  // the fallback path redoes the frame
    const uint32_t split = A.f->s_counts[SC_SHADOW_SPLIT];
    const uint32_t first = pass == 0 ? 0u : split;
    const uint32_t total = pass == 0 ? split : A.f->s_counts[SC_SHADOW] - split;
    uint32_t *cursor = A.f->s_counts + (pass == 0 ? SC_SHADOW_FETCH : SC_SHADOW_FETCH2);
    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;

    Ray R;
    LaneWalk L;
    int state = ST_FETCH;
    uint32_t r = 0, steps = 0;
    for (;;) {
        if (__ballot(state == ST_FETCH)) {
            if (state == ST_FETCH) {
                r = wave_fetch(cursor, lane);
                if (r >= total) state = ST_DONE;
                else {
                    r += first;
                    const float4 q0 = A.f->s_shadowq[2 * (size_t)r], q1 = A.f->s_shadowq[2 * (size_t)r + 1];
                    if (__float_as_uint(q0.w) != SHADOW_SLOT_UNUSED) {  // (unused: the lane fetches again next trip)
                        R.ox = q0.x; R.oy = q0.y; R.oz = q0.z;
                        R.dx = q1.x; R.dy = q1.y; R.dz = q1.z;  // already normalised once; shadow rays skip shootRay (RayTracer.cpp:313-317)
                        ray_prepare(R);
                        L.rtype = RAY_SHADOW;
                        traversal_begin(L, A.s->top_root);
                        L.light_dist = q0.w;
                        state = ST_TRAVERSE;
                    }
                }
            }
        }
        if (!__ballot(state != ST_DONE)) break;
        if (state == ST_TRAVERSE) {
            int budget = 64;
            bool finished = false;
            do {
                if (!traversal_step<COUNT>(L, R, A, cnt)) { finished = true; break; }
                steps++;
            } while (--budget > 0);
            if (finished) {
                A.f->s_occluded[r] = L.occluded ? 1 : 0;
                steps = 0;
                state = ST_FETCH;
            } else if (A.step_budget && steps >= A.step_budget) {  // a long walk: hand it to heavy_trace_shadow
                const unsigned long long em = __ballot(1);
                uint32_t base = 0;
                if ((em & ((1ull << lane) - 1ull)) == 0) base = atomicAdd(A.f->s_counts + SC_SHEAVY, (uint32_t)__popcll(em));
                base = __shfl(base, __ffsll((long long)em) - 1);
                const uint32_t slot = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
                if (slot < A.f->s_heavy_cap) {
                    A.f->s_sheavy[slot] = r;
                    state = ST_FETCH;
                }
                steps = 0;
            }
        }
    }
    if (COUNT) {
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters[k], v);
        }
    }
}

// Colour of a finished leaf of the ray tree.
template <bool COUNT>
__device__ __forceinline__ void resolve_leaf(const KernelArgs &A, const float4 &n0, const float4 &n1, float &cx, float &cy,
                                             float &cz, uint32_t *cnt) {
    const uint32_t kind = __float_as_uint(n0.w);
    if ((kind & TN_KIND_MASK) == TN_DIFFUSE) {
        // finalColor += (intensity / area * angle) * albedo, light by light in the reference's order (RayTracer.cpp:319-329)
        const uint32_t first = __float_as_uint(n1.x), stride = __float_as_uint(n1.y);
        float ax = 0, ay = 0, az = 0;
        for (uint32_t li = 0; li < A.s->n_lights; li++) {
            const size_t slot = (size_t)first + (size_t)li * stride;
            if (!A.f->s_occluded[slot]) {
                const float k = A.f->s_shadowq[2 * slot + 1].w;
                if (COUNT && (kind & TN_BITMAP)) cnt[C_TEXEL]++;
                ax += k * n0.x; ay += k * n0.y; az += k * n0.z;
            }
        }
        cx = ax; cy = ay; cz = az;
    } else {
        cx = n0.x; cy = n0.y; cz = n0.z;  // TN_CONST
    }
}

// Per pixel: evaluate its ray tree in post-order -- reflection subtree, then refraction subtree, then the
// parent's own expression; a GI diffuse node: its direct light, then its samples' subtrees one after the other -- exactly the
// order of the reference's recursion.  GI mode: the pixel's level0_samples trees, summed in order and averaged (RayTracer.cpp:90-104).
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void stream_resolve(const KernelArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.f->s_counts[SC_OVERFLOW]) return;  // the fallback path redoes the frame
    const uint32_t total = A.f->n_items * 64u;
    const float4 *nodes = A.f->s_nodes;
    float4 *wnodes = A.f->s_nodes;
    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;
    uint32_t stack[MAX_GENERATIONS];
    const bool gi_mode = A.f->use_gi != 0;
    const float gi_inv = 1.0f / (float)(A.f->gi_samples + 1u);  // RayTracer.cpp:352-353

    for (uint32_t q = blockIdx.x * BLOCK + threadIdx.x; q < total; q += gridDim.x * BLOCK) {
        if ((__float_as_uint(nodes[2 * (size_t)q].w) & TN_KIND_MASK) == TN_SKIP) continue;  // (coverage is the same for every sample)
        float sumx = 0, sumy = 0, sumz = 0;
        float cx = 0, cy = 0, cz = 0;
        for (uint32_t sample = 0; sample < A.f->level0_samples; sample++) {
            int sp = 0;
            uint32_t cur = sample * total + q;
            bool descending = true;
            // iterative post-order: `descending` = entering node `cur`; otherwise (cx,cy,cz) is the value returned
            // by the child just finished and stack[sp-1] is its parent.
            for (;;) {
                if (descending) {
                    if (cur == CHILD_BG) { cx = A.s->bgx; cy = A.s->bgy; cz = A.s->bgz; descending = false; }
                    else {
                        const float4 n0 = nodes[2 * (size_t)cur], n1 = nodes[2 * (size_t)cur + 1];
                        const uint32_t kind = __float_as_uint(n0.w) & TN_KIND_MASK;
                        if (kind == TN_REFLECT || kind == TN_REFRACT) {
                            stack[sp++] = cur;
                            cur = __float_as_uint(n1.x);  // the reflection ray is shot first (RayTracer.cpp:366, 398-400)
                        } else {
                            resolve_leaf<COUNT>(A, n0, n1, cx, cy, cz, cnt);
                            if (__float_as_uint(n0.w) & TN_GI) {
                                if (A.f->gi_samples == 0u) {  // finalColor += (0, 0, 0); return finalColor * (1 / 1)
                                    cx = (cx + 0.0f) * gi_inv; cy = (cy + 0.0f) * gi_inv; cz = (cz + 0.0f) * gi_inv;
                                    descending = false;
                                } else {
                                    // park the direct light and an empty indirect sum in the node, descend into sample 0
                                    const uint32_t first = __float_as_uint(n1.z);
                                    wnodes[2 * (size_t)cur] = make_float4(cx, cy, cz, __uint_as_float((__float_as_uint(n0.w) & 0xFFFFu) | TN_PARKED));
                                    wnodes[2 * (size_t)cur + 1] = make_float4(0.0f, 0.0f, n1.z, 0.0f);
                                    stack[sp++] = cur;
                                    cur = first;
                                }
                            } else descending = false;
                        }
                    }
                }
                if (!descending) {
                    if (sp == 0) break;
                    const uint32_t p = stack[sp - 1];
                    const float4 p0 = nodes[2 * (size_t)p], p1 = nodes[2 * (size_t)p + 1];
                    const uint32_t kind = __float_as_uint(p0.w);
                    if ((kind & TN_KIND_MASK) == TN_DIFFUSE) {
                        // indirectLightContribution += shootRay(sample i) (RayTracer.cpp:350); {x, y, w} of the second word hold the sum
                        const float ix = p1.x + cx, iy = p1.y + cy, iz = p1.w + cz;
                        const uint32_t done = (kind >> TN_GI_DONE_SHIFT) + 1u;
                        if (done < A.f->gi_samples) {
                            wnodes[2 * (size_t)p] = make_float4(p0.x, p0.y, p0.z, __uint_as_float((kind & 0xFFFFu) | (done << TN_GI_DONE_SHIFT)));
                            wnodes[2 * (size_t)p + 1] = make_float4(ix, iy, p1.z, iz);
                            const uint32_t first = __float_as_uint(p1.z);
                            cur = first == CHILD_BG ? CHILD_BG : first + done;
                            descending = true;
                        } else {
                            cx = (p0.x + ix) * gi_inv; cy = (p0.y + iy) * gi_inv; cz = (p0.z + iz) * gi_inv;  // RayTracer.cpp:351-353
                            sp--;
                        }
                    } else if ((kind & TN_KIND_MASK) == TN_REFLECT) {
                        cx = 0.0f + p0.x * cx; cy = 0.0f + p0.y * cy; cz = 0.0f + p0.z * cz;  // RayTracer.cpp:368-372
                        sp--;
                    } else if (kind & TN_PARKED) {
                        // both children done: fresnel * reflection + (1 - fresnel) * refraction (RayTracer.cpp:414)
                        const float f = p1.z;
                        cx = f * p0.x + (1 - f) * cx; cy = f * p0.y + (1 - f) * cy; cz = f * p0.z + (1 - f) * cz;
                        sp--;
                    } else {
                        const uint32_t refr = __float_as_uint(p1.y);
                        if (refr == CHILD_NONE) { sp--; }  // `return reflectionColor`, RayTracer.cpp:416
                        else {
                            // park the reflection colour in the node, mark it, descend into the refraction child
                            wnodes[2 * (size_t)p] = make_float4(cx, cy, cz, __uint_as_float(kind | TN_PARKED));
                            cur = refr;
                            descending = true;
                        }
                    }
                }
            }
            if (gi_mode) { sumx = sumx + cx; sumy = sumy + cy; sumz = sumz + cz; }  // std::accumulate from Color(0, 0, 0), RayTracer.cpp:101
        }
        if (gi_mode) {
            const float inv = 1.0f / (float)A.f->level0_samples;
            cx = sumx * inv; cy = sumy * inv; cz = sumz * inv;
        }
        const WorkItem wi = A.f->items[q >> 6];
        const uint32_t sub = q & 63u;
        const uint32_t px = (wi.tile % A.s->tiles_x) * TILE + (sub & 7u), py = (wi.tile / A.s->tiles_x) * TILE + (sub >> 3);
        const size_t out_off = A.f->packed ? ((size_t)wi.out_tile * 64 + sub) * 3 : ((size_t)py * A.s->width + px) * 3;
        A.f->out[out_off] = cx; A.f->out[out_off + 1] = cy; A.f->out[out_off + 2] = cz;  // RayTracer.cpp:106
    }
    if (COUNT) {
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters[k], v);
        }
    }
}
