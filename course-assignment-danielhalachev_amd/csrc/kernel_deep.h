// kernel_deep.h -- the recursion levels >= 1 as ONE persistent, queue-driven launch.
//
// Unrolled level by level (crt_device.hip: one launch triple per level) the deeper levels are nine dependent launches of
// a few ten thousand rays each: every level waits for its longest walk, then for a launch boundary.  The levels are not
// a dependency of the ALGORITHM, though -- only a ray's own ancestors are.  Here every wave runs
//
//     loop:  claim the next slot of the ray queue  ->  wait until its record is published (or nothing can come any more)
//            -> walk the ray (kernel_heavy.h: one ray per wave)  ->  shade the hit  ->  the reflection ray, if any, is
//            walked next by this very wave; the transmission ray, if any, is published into the queue
//
// with no barrier between levels.  The ray tree is the same as in the per-level form -- a child's node index is allocated
// when the child is created and travels with the ray -- and stream_resolve evaluates it in the reference's post-order, so
// the order in which the waves happen to produce the nodes cannot change a bit of the frame.
//
// The launch takes over at level f = KernelArgs::deep_first (1: everything below the primary rays; 3: only the thin levels,
// whose launches are all latency -- crt_tuning::deep_first).  Queue = s_rayq[f & 1].  Slots [0, tail0) are the level-f rays
// that level f - 1 left there (complete before this launch).
//   tail   s_counts[SC_COUNT + f]   slots reserved so far (producers add to it BEFORE they bump `done`)
//   head   s_counts[SC_FETCH + f]   slots claimed so far (a claim may run ahead of tail: that wave waits for its slot)
//   done   s_counts[SC_DEEP_DONE]   claimed slots whose whole chain has been finished
// Publication of a slot (agent scope, MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility"): EVERY
// dword of the record is written with a relaxed agent-scope atomic store (global_store sc1: written through, coherent
// per location by the memory model) -> s_waitcnt vmcnt(0) (the stores have completed) -> relaxed agent-scope store of the
// frame's epoch to s_ready[slot]; the consumer polls s_ready[slot] with relaxed agent-scope loads and, once it has
// matched, reads EVERY dword of the record with agent-scope loads.  No fences: an agent-scope release is a write-back of
// the XCD's whole L2 (buffer_wbl2), and one per published ray made this launch 3x slower than the per-level form
// (measured: 17 ms against 5.3 ms for the levels of the benchmark frame).
// Termination: a waiting wave leaves when done == tail (read tail, done, tail: equal and unchanged) -- every reserved
// slot is finished, so no wave is left that could reserve another -- or when the overflow word is up.  Every wait is
// bounded (DEEP_SPIN_LIMIT polls, then the overflow word is raised and the frame is redone by render_lanes), and no wave
// ever waits for a wave that is not running: whoever holds unfinished work is resident and finishes it on its own.
#pragma once

#include "kernel_common.h"
#include "kernel_heavy.h"
#include "kernel_stream.h"

constexpr uint32_t DEEP_SPIN_LIMIT = 1u << 21;   // polls of ~1 us each: seconds, never reached by a healthy frame

__device__ __forceinline__ uint32_t agent_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float agent_loadf(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void agent_storef(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread, between level 0 and the deep launch
__global__ void deep_begin(const KernelArgs A) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t first = A.s_counts[SC_COUNT + A.deep_first];
    A.s_counts[SC_DEEP_TAIL0] = first;
    A.s_counts[SC_DEEP_NODES] = stream_level_base(A, A.deep_first) + first;   // ray k of level g owns node base(g) + k (shade_and_emit)
    A.s_counts[SC_DEEP_DONE] = 0;
    A.s_counts[SC_DEEP_CHAINED] = 0;
}

constexpr uint32_t DEEP_NODE_CHUNK = 32;   // ray-tree nodes a wave takes from the allocation cursor at a time
constexpr uint32_t DEEP_PREPARED = 0x80000000u;  // heavy-queue records: the direction has already been normalised at shootRay's entry

// One record of a deep queue, written so that another wave may read it: every dword an agent-scope store.
__device__ __forceinline__ void deep_store_record(float4 *q, const size_t index, const float ox, const float oy, const float oz,
                                                  const uint32_t level, const float dx, const float dy, const float dz, const uint32_t node) {
    float *rec = reinterpret_cast<float *>(q + 2 * index);
    agent_storef(rec + 0, ox); agent_storef(rec + 1, oy); agent_storef(rec + 2, oz); agent_storef(rec + 3, __uint_as_float(level));
    agent_storef(rec + 4, dx); agent_storef(rec + 5, dy); agent_storef(rec + 6, dz); agent_storef(rec + 7, __uint_as_float(node));
}

// A ray and, for as long as its hit is a mirror, its reflection rays, walked by the whole wave (kernel_heavy.h); the
// transmission rays go into the queue.  Returns false when a queue ran full (the overflow word is up).
__device__ __forceinline__ bool deep_wave_chain(const KernelArgs &A, const TopRegs &TR, Ray R, uint32_t gen, uint32_t node, bool prepared,
                                                const uint32_t lane, uint32_t &node_next, uint32_t &node_left, uint32_t &chained) {
    uint32_t *const tail = A.s_counts + SC_COUNT + A.deep_first;
    for (uint32_t hops = 0; hops <= A.max_depth; hops++) {  // (a chain cannot be longer than the recursion is deep)
        if (!prepared) normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
        prepared = false;
        ray_prepare(R);
        R.ox = uniform_f(R.ox); R.oy = uniform_f(R.oy); R.oz = uniform_f(R.oz);
        R.dx = uniform_f(R.dx); R.dy = uniform_f(R.dy); R.dz = uniform_f(R.dz);
        R.ix = uniform_f(R.ix); R.iy = uniform_f(R.iy); R.iz = uniform_f(R.iz);
        R.parmask = __builtin_amdgcn_readfirstlane(R.parmask);
        gen = __builtin_amdgcn_readfirstlane(gen);
        node = __builtin_amdgcn_readfirstlane(node);
        bool have = false, occluded = false;
        float bt = 0;
        uint32_t btri = 0, bmesh = 0;
        heavy_walk<false>(A, TR, R, false, 0.0f, have, bt, btri, bmesh, occluded, lane);
        // shootRay's dispatch, by one lane (the hit is wave-uniform); lane 0 then tells the wave how to go on
        uint32_t go = 0, next_node = 0;
        float nox = 0, noy = 0, noz = 0, ndx = 0, ndy = 0, ndz = 0;
        if (lane == 0) {
            Shaded E;
            shade_hit<false>(A, gen, 0u, R, have, bt, btri, bmesh, nullptr, lane, E);
            if (E.reflect) {
                const uint32_t n = E.transmit ? 2u : 1u;
                if (node_left < n) { node_next = atomicAdd(A.s_counts + SC_DEEP_NODES, DEEP_NODE_CHUNK); node_left = DEEP_NODE_CHUNK; }
                const uint32_t nb = node_next;
                node_next += n; node_left -= n;
                uint32_t ts = 0;
                if (E.transmit) ts = atomicAdd(tail, 1u);
                if ((uint64_t)nb + n > A.s_node_cap || (E.transmit && ts >= A.s_ray_cap)) {
                    atomicExch(A.s_counts + SC_OVERFLOW, 1u);
                    go = 2;  // stop: the frame is redone by the fallback
                } else {
                    E.N.a = nb;
                    if (E.transmit) {
                        E.N.b = nb + 1u;
                        deep_store_record(A.s_rayq[A.deep_first & 1u], ts, E.tox, E.toy, E.toz, gen + 1u, E.tdx, E.tdy, E.tdz, nb + 1u);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record is in place before the flag goes up
                        __hip_atomic_store(A.s_ready + ts, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    go = 1;
                    next_node = nb;
                    nox = E.rox; noy = E.roy; noz = E.roz; ndx = E.rdx; ndy = E.rdy; ndz = E.rdz;
                }
            }
            store_tnode(A, (size_t)node, E.N);
        }
        go = __builtin_amdgcn_readfirstlane(go);
        if (go != 1u) return go != 2u;
        R.ox = uniform_f(nox); R.oy = uniform_f(noy); R.oz = uniform_f(noz);
        R.dx = uniform_f(ndx); R.dy = uniform_f(ndy); R.dz = uniform_f(ndz);
        node = __builtin_amdgcn_readfirstlane(next_node);
        gen = gen + 1u;
        chained++;
    }
    return true;
}

// Reads a queue record another wave may have written (agent-scope loads); false when it cannot be one of this frame's.
__device__ __forceinline__ bool deep_load_record(const KernelArgs &A, const float4 *q, const size_t index, Ray &R, uint32_t &gen, uint32_t &node) {
    const float *rec = reinterpret_cast<const float *>(q + 2 * index);
    R.ox = agent_loadf(rec + 0); R.oy = agent_loadf(rec + 1); R.oz = agent_loadf(rec + 2);
    gen = __float_as_uint(agent_loadf(rec + 3));
    R.dx = agent_loadf(rec + 4); R.dy = agent_loadf(rec + 5); R.dz = agent_loadf(rec + 6);
    node = __float_as_uint(agent_loadf(rec + 7));
    const uint32_t level = gen & ~DEEP_PREPARED;
    return node < A.s_node_cap && level != 0u && level <= A.max_depth;
}

// "Everything that was ever reserved has been finished": tail, done, tail read in this order, equal and unchanged.
__device__ __forceinline__ bool deep_all_done(const KernelArgs &A, uint32_t &tail_now) {
    const uint32_t t1 = agent_load(A.s_counts + SC_COUNT + A.deep_first), d = agent_load(A.s_counts + SC_DEEP_DONE), t2 = agent_load(A.s_counts + SC_COUNT + A.deep_first);
    tail_now = t1;
    return t1 == t2 && d == t1;
}

template <int WAVES_PER_SIMD>  // register budget: 4 = what the compiler takes by itself (104 VGPRs), 5 = 96 VGPRs and a few spills in the shading
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void deep_trace(const KernelArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.s_counts[SC_OVERFLOW]) return;
    uint32_t *const head = A.s_counts + SC_FETCH + A.deep_first;
    uint32_t *const done = A.s_counts + SC_DEEP_DONE;
    const uint32_t tail0 = A.s_counts[SC_DEEP_TAIL0];   // written by deep_begin, a launch ago
    const TopRegs TR = heavy_top_load(A, lane);
    // wave-uniform bookkeeping.  Counters that every ray would otherwise hit with an atomic are touched rarely: finished
    // slots are reported when the wave is about to wait or to leave (nobody can conclude "all done" while this wave still
    // holds unreported work, and a wave that waits has reported); ray-tree nodes are taken DEEP_NODE_CHUNK at a time.
    uint32_t chained = 0, finished = 0, node_next = 0, node_left = 0, waits = 0;

    for (uint32_t claims = 0;; claims++) {
        if ((claims & 15u) == 15u && agent_load(A.s_counts + SC_OVERFLOW)) break;  // a queue ran full somewhere: the fallback redoes the frame
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(head, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= tail0) {
            // ---- wait for the slot's record, or for the end of the work
            bool alive = true;
            uint32_t spins = 0;
            for (;;) {
                const uint32_t ready = slot < A.s_ray_cap ? agent_load(A.s_ready + slot) : 0u;
                if (ready == A.epoch) break;
                if (finished) {  // (before the first look at the counters: they must include this wave's own work)
                    if (lane == 0) atomicAdd(done, finished);
                    finished = 0;
                }
                if ((spins & 7u) == 0u) {
                    uint32_t t = 0;
                    if ((deep_all_done(A, t) && slot >= t) || agent_load(A.s_counts + SC_OVERFLOW)) { alive = false; break; }
                }
                if (++spins > DEEP_SPIN_LIMIT) {  // cannot happen unless a publication was lost: give the frame to the fallback
                    if (lane == 0) { A.s_counts[SC_GUARD] = 2; atomicExch(A.s_counts + SC_OVERFLOW, 1u); }
                    alive = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            waits += spins ? 1u : 0u;
            if (!alive) break;
        }
        // ---- the slot's ray; then, for as long as a hit is a mirror, its reflection ray without going through the queue
        Ray R;
        uint32_t gen = 0, node = 0;
        if (!deep_load_record(A, A.s_rayq[A.deep_first & 1u], slot, R, gen, node)) {  // not a record of this frame: never follow it
            if (lane == 0) { A.s_counts[SC_GUARD] = 3; atomicExch(A.s_counts + SC_OVERFLOW, 1u); }
            break;
        }
        const bool ok = deep_wave_chain(A, TR, R, gen, node, false, lane, node_next, node_left, chained);
        finished++;
        if (!ok) break;
    }
    if (lane == 0) {
        if (finished) atomicAdd(done, finished);
        if (chained) atomicAdd(A.s_counts + SC_DEEP_CHAINED, chained);  // diagnostics: rays that never saw the queue,
        if (waits) atomicAdd(A.s_counts + SC_DEEP_WAITS, waits);        // claims that had to wait for their record
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// deep_lanes: the same level-free queue, but ONE RAY PER LANE for the common case.
//
// A wave walking one ray (deep_trace) answers fast and is expensive: ~110 wave-level loads and ~2,400 vector instructions
// per ray, a wave slot for ~40 us.  The per-lane walk of kernel_plan.h costs a tenth of that per ray and takes longer to
// answer -- which does not matter as long as the whole chain of a pixel fits inside the bulk shadow pass running beside
// this launch.  So the waves of this kernel come in two roles, fixed by their index in the workgroup:
//   lane waves   every lane claims a queue slot, waits (without blocking the other lanes) until its record is published,
//                plans and walks the ray (binary nodes), shades it, publishes the transmission child, goes on with the
//                reflection child itself; a walk that outlasts the step budget, or a ray with a direction component below
//                FLT_EPSILON, is handed over through the heavy queue (the other s_rayq, flags s_ready2);
//   heavy waves  (every A.deep_heavy_every-th wave) take those rays and finish their chains the way deep_trace does.
// Every workgroup holds both roles, so whatever is resident can finish whatever exists: no wave ever waits for work that
// only a non-resident wave could do.  Accounting as in deep_trace: a claimed slot is `done` when its chain has ended --
// reported by the lane wave, or by the heavy wave that took the chain over.
#include "kernel_plan.h"

enum : int { ST_PENDING = 3 };  // (beside ST_FETCH / ST_TRAVERSE / ST_DONE) a claimed slot whose record is not there yet

__device__ __forceinline__ void deep_heavy_role(const KernelArgs &A, const uint32_t lane) {
    uint32_t *const hq_head = A.s_counts + SC_DEEP_HQ_HEAD;
    uint32_t *const done = A.s_counts + SC_DEEP_DONE;
    const TopRegs TR = heavy_top_load(A, lane);
    uint32_t chained = 0, node_next = 0, node_left = 0;
    for (;;) {
        uint32_t h = 0;
        if (lane == 0) h = atomicAdd(hq_head, 1u);
        h = __builtin_amdgcn_readfirstlane(h);
        bool alive = true;
        for (uint32_t spins = 0;; spins++) {
            const uint32_t ready = h < A.s_ray_cap ? agent_load(A.s_ready2 + h) : 0u;
            if (ready == A.epoch) break;
            // no record: a ray in the heavy queue belongs to a slot that is not done, so "all done" means none will come
            uint32_t t = 0;
            if ((spins & 3u) == 0u && (deep_all_done(A, t) || agent_load(A.s_counts + SC_OVERFLOW))) { alive = false; break; }
            if (spins > DEEP_SPIN_LIMIT) {
                if (lane == 0) { A.s_counts[SC_GUARD] = 4; atomicExch(A.s_counts + SC_OVERFLOW, 1u); }
                alive = false;
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        if (!alive) break;
        Ray R;
        uint32_t gen = 0, node = 0;
        if (!deep_load_record(A, A.s_rayq[(A.deep_first & 1u) ^ 1u], h, R, gen, node)) {
            if (lane == 0) { A.s_counts[SC_GUARD] = 5; atomicExch(A.s_counts + SC_OVERFLOW, 1u); }
            break;
        }
        const bool ok = deep_wave_chain(A, TR, R, gen & ~DEEP_PREPARED, node, (gen & DEEP_PREPARED) != 0u, lane, node_next, node_left, chained);
        if (lane == 0) atomicAdd(done, 1u);
        if (!ok) break;
    }
    if (lane == 0 && chained) atomicAdd(A.s_counts + SC_DEEP_CHAINED, chained);
}

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void deep_lanes(const KernelArgs A) {
    extern __shared__ uint32_t plan_lds[];  // mesh lists: A.plan_list_words x BLOCK (kernel_plan.h)
    __shared__ TopLdsStorage top_storage;
    const TopLds TL = top_lds_load(A, top_storage);  // (a barrier inside: before any return; none after it)
    const uint32_t lane = threadIdx.x & 63u;
    if (A.s_counts[SC_OVERFLOW]) return;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (A.deep_heavy_every && (wave_in_block % A.deep_heavy_every) == A.deep_heavy_every - 1u) { deep_heavy_role(A, lane); return; }

    uint32_t *const head = A.s_counts + SC_FETCH + A.deep_first;
    uint32_t *const tail = A.s_counts + SC_COUNT + A.deep_first;
    uint32_t *const done = A.s_counts + SC_DEEP_DONE;
    const uint32_t tail0 = A.s_counts[SC_DEEP_TAIL0];
    const char *nodes_b = reinterpret_cast<const char *>(A.pnodes);
    const char *ptris_b = reinterpret_cast<const char *>(A.ptris);
    PlanList PL;
    PL.words = plan_lds + threadIdx.x;
    PL.count = 0; PL.next = 0;

    Ray R;
    uint32_t gen = 0, node = 0, slot = 0;
    uint32_t wn = END, we = NONE, mesh = NONE, mtri = 0, btri = 0, bmesh = 0;
    float mmin = INFINITY, mt = 0, tmin = INFINITY, bt = 0;
    bool mhave = false, have = false;
    bool chain_fresh = false;      // the lane goes on with the reflection child it has just made
    uint32_t nbox = 0, ntri = 0, nplan = 0, steps = 0;
    uint32_t finished = 0;         // wave-uniform: chains ended here and not yet added to `done`
    int state = ST_FETCH;
    for (uint32_t trip = 0;; trip++) {
        // ---- 1. free lanes claim queue slots, one atomic for all of them; a claim may run ahead of the queue's tail
        if (__ballot(state == ST_FETCH) && (A.bundle >= 64u || (uint32_t)__popcll(__ballot(state == ST_TRAVERSE)) <= A.bundle)) {
            if (state == ST_FETCH) { slot = wave_fetch(head, lane); state = ST_PENDING; }
        }
        // ---- 2. claimed slots: is the record there?
        bool fresh = chain_fresh;
        chain_fresh = false;
        if (state == ST_PENDING) {
            const bool ready = slot < tail0 || (slot < A.s_ray_cap && agent_load(A.s_ready + slot) == A.epoch);
            if (ready) {
                if (deep_load_record(A, A.s_rayq[A.deep_first & 1u], slot, R, gen, node)) {
                    normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                    fresh = true;
                    state = ST_TRAVERSE;
                } else {  // not a record of this frame: never follow it; the fallback redoes the frame
                    A.s_counts[SC_GUARD] = 6;
                    atomicExch(A.s_counts + SC_OVERFLOW, 1u);
                    state = ST_DONE;
                }
            }
        }
        // ---- 3. new rays: the plan (wave-uniform loop over the top-level leaves), or the way to a heavy wave
        bool to_heavy = false;
        if (fresh) {
            ray_prepare(R);
            if (R.parmask != 0) to_heavy = true;  // BoundingBox.h:90-93 needs the general box test, which the wave-per-ray walk has
            else {
                plan_closest_meshes(A, R, PL, nplan);
                wn = END; we = NONE; mesh = NONE;
                mhave = false; mmin = INFINITY; mt = 0; mtri = 0;
                have = false; tmin = INFINITY; bt = 0; btri = 0; bmesh = 0;
                steps = 0;
            }
        }
        // ---- 4. walk (kernel_plan.h: stream_trace_shade_plan<false>, the binary-node form)
        bool walked = false;
        if (state == ST_TRAVERSE && !to_heavy) {
            for (int it = 0; it < 64; ++it) {
                steps++;
                if (we != NONE) {
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
                    // deeper rays are never culled (Ray.cpp:13 is PrimaryRay only)
                    const bool ok = !(t < 0) && !(s0 < -FLT_EPSILON) && !(s1 < -FLT_EPSILON) && !(s2 < -FLT_EPSILON);
                    const bool less = ok && (t < mmin);  // KDTree.cpp:75-86
                    const bool take = less || (ok && !mhave);
                    mt = take ? t : mt;
                    mtri = take ? entry : mtri;
                    mmin = less ? t : mmin;
                    mhave = mhave || ok;
                    we = leaf_cursor_next(we);
                } else {
                    if (wn == END) {
                        // a mesh ended (scene-level rule, KDTree.cpp:156-167), the next one begins
                        if (mesh != NONE && mhave) {
                            if (!have) { have = true; bt = mt; btri = mtri; bmesh = mesh; }
                            if (mt < tmin) { tmin = mt; bt = mt; btri = mtri; bmesh = mesh; }
                        }
                        if (PL.next >= PL.count) { walked = true; break; }
                        mesh = plan_list_pop(PL);
                        wn = TL.meshes[mesh][1];
                        mhave = false;
                        mmin = INFINITY;
                    }
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
            if (!walked && steps >= A.step_budget) to_heavy = true;  // a long walk: a whole wave finishes it faster
        }
        // ---- 5. shootRay's dispatch for the walks that ended; children
        bool published = false, chain_end = false;
        uint32_t pub_slot = 0;
        if (walked) {
            if (have) btri = A.leaf_tris[btri] & ~LAST;  // leaf entry -> triangle
            Shaded E;
            shade_hit<false>(A, gen, 0u, R, have, bt, btri, bmesh, nullptr, lane, E);
            bool go_on = false;
            if (E.reflect) {
                // ray-tree nodes and queue slots for the children, one atomic each for all lanes here
                const unsigned long long m1 = __ballot(1), m2 = __ballot(E.transmit);
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
                uint32_t nbase = 0, tbase = 0;
                if ((m1 & below) == 0) { nbase = atomicAdd(A.s_counts + SC_DEEP_NODES, n1 + n2); if (n2) tbase = atomicAdd(tail, n2); }
                nbase = __shfl(nbase, __ffsll((long long)m1) - 1);
                tbase = __shfl(tbase, __ffsll((long long)m1) - 1);
                if ((uint64_t)nbase + n1 + n2 > A.s_node_cap || (uint64_t)tbase + n2 > A.s_ray_cap) {
                    atomicExch(A.s_counts + SC_OVERFLOW, 1u);
                } else {
                    const uint32_t my_node = nbase + (uint32_t)__popcll(m1 & below);
                    E.N.a = my_node;
                    if (E.transmit) {
                        const uint32_t k = (uint32_t)__popcll(m2 & below);
                        E.N.b = nbase + n1 + k;
                        pub_slot = tbase + k;
                        deep_store_record(A.s_rayq[A.deep_first & 1u], pub_slot, E.tox, E.toy, E.toz, gen + 1u, E.tdx, E.tdy, E.tdz, nbase + n1 + k);
                        published = true;
                    }
                    go_on = true;
                }
            }
            store_tnode(A, (size_t)node, E.N);
            if (go_on) {  // the reflection child, by this lane, now
                R.ox = E.rox; R.oy = E.roy; R.oz = E.roz;
                R.dx = E.rdx; R.dy = E.rdy; R.dz = E.rdz;
                normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
                node = E.N.a;
                gen = gen + 1u;
                chain_fresh = true;
            } else {
                chain_end = true;
                state = ST_FETCH;
            }
        }
        // ---- 6. rays for the heavy waves
        uint32_t hq_slot = 0;
        if (to_heavy) {
            const unsigned long long m = __ballot(1);
            const unsigned long long below = (1ull << lane) - 1ull;
            uint32_t hbase = 0;
            if ((m & below) == 0) hbase = atomicAdd(A.s_counts + SC_DEEP_HQ_TAIL, (uint32_t)__popcll(m));
            hbase = __shfl(hbase, __ffsll((long long)m) - 1);
            hq_slot = hbase + (uint32_t)__popcll(m & below);
            if (hq_slot >= A.s_ray_cap) { atomicExch(A.s_counts + SC_OVERFLOW, 1u); to_heavy = false; }
            else deep_store_record(A.s_rayq[(A.deep_first & 1u) ^ 1u], hq_slot, R.ox, R.oy, R.oz, gen | DEEP_PREPARED, R.dx, R.dy, R.dz, node);
            state = ST_FETCH;  // the slot's `done` is the heavy wave's to report
            chain_fresh = false;
        }
        // ---- 7. flags after the records (every storing lane's stores have completed: one wait for the wave)
        if (__ballot(published || to_heavy)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (published) __hip_atomic_store(A.s_ready + pub_slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (to_heavy) __hip_atomic_store(A.s_ready2 + hq_slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        finished += (uint32_t)__popcll(__ballot(chain_end));
        // ---- 8. waiting lanes: report, look for the end of the work
        const unsigned long long pending = __ballot(state == ST_PENDING);
        const unsigned long long busy = __ballot(state == ST_TRAVERSE || state == ST_FETCH);
        if (pending) {
            if (finished) { if (lane == 0) atomicAdd(done, finished); finished = 0; }
            if (!busy || (trip & 3u) == 0u) {
                uint32_t t = 0;
                const bool all_done = deep_all_done(A, t);
                const bool stop = agent_load(A.s_counts + SC_OVERFLOW) != 0u;
                if (state == ST_PENDING && ((all_done && slot >= t) || stop)) state = ST_DONE;
                if (stop && state != ST_DONE) state = ST_DONE;  // a queue ran full: the fallback redoes the frame
            }
            if (!busy) {
                if (trip > DEEP_SPIN_LIMIT) { if (lane == 0) { A.s_counts[SC_GUARD] = 7; atomicExch(A.s_counts + SC_OVERFLOW, 1u); } break; }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        if (!__ballot(state != ST_DONE)) break;
    }
    if (lane == 0 && finished) atomicAdd(done, finished);
    exec_counters_flush(A, nbox, ntri, lane, nplan);
}
