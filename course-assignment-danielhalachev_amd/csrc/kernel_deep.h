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
// Queue = s_rayq[1].  Slots [0, tail0) are the level-1 rays that level 0 left there (complete before this launch).
//   tail   s_counts[SC_COUNT + 1]   slots reserved so far (producers add to it BEFORE they bump `done`)
//   head   s_counts[SC_FETCH + 1]   slots claimed so far (a claim may run ahead of tail: that wave waits for its slot)
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
    const uint32_t level1 = A.s_counts[SC_COUNT + 1];
    A.s_counts[SC_DEEP_TAIL0] = level1;
    A.s_counts[SC_DEEP_NODES] = A.n_items * 64u + level1;   // level-1 ray k owns node n_items * 64 + k (shade_and_emit)
    A.s_counts[SC_DEEP_DONE] = 0;
    A.s_counts[SC_DEEP_CHAINED] = 0;
}

constexpr uint32_t DEEP_NODE_CHUNK = 32;   // ray-tree nodes a wave takes from the allocation cursor at a time

template <int WAVES_PER_SIMD>  // register budget: 4 = what the compiler takes by itself (104 VGPRs), 5 = 96 VGPRs and a few spills in the shading
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void deep_trace(const KernelArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    if (A.s_counts[SC_OVERFLOW]) return;
    uint32_t *const head = A.s_counts + SC_FETCH + 1;
    uint32_t *const tail = A.s_counts + SC_COUNT + 1;
    uint32_t *const done = A.s_counts + SC_DEEP_DONE;
    const uint32_t tail0 = A.s_counts[SC_DEEP_TAIL0];   // written by deep_begin, a launch ago
    const float4 *const q = A.s_rayq[1];
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
                    const uint32_t t1 = agent_load(tail), d = agent_load(done), t2 = agent_load(tail);
                    if ((t1 == t2 && d == t1 && slot >= t1) || agent_load(A.s_counts + SC_OVERFLOW)) { alive = false; break; }
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
        const float *rec = reinterpret_cast<const float *>(q + 2 * (size_t)slot);
        Ray R;
        R.ox = agent_loadf(rec + 0); R.oy = agent_loadf(rec + 1); R.oz = agent_loadf(rec + 2);
        uint32_t gen = __float_as_uint(agent_loadf(rec + 3));
        R.dx = agent_loadf(rec + 4); R.dy = agent_loadf(rec + 5); R.dz = agent_loadf(rec + 6);
        uint32_t node = __float_as_uint(agent_loadf(rec + 7));
        bool overflowed = false;
        if (node >= A.s_node_cap || gen == 0u || gen > A.max_depth) {  // not a record of this frame: never follow it
            if (lane == 0) { A.s_counts[SC_GUARD] = 3; atomicExch(A.s_counts + SC_OVERFLOW, 1u); }
            break;
        }
        for (uint32_t hops = 0; hops <= A.max_depth; hops++) {  // (a chain cannot be longer than the recursion is deep)
            normalize3(R.dx, R.dy, R.dz);  // shootRay entry (RayTracer.cpp:420)
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
                            float *rec = reinterpret_cast<float *>(A.s_rayq[1] + 2 * (size_t)ts);
                            agent_storef(rec + 0, E.tox); agent_storef(rec + 1, E.toy); agent_storef(rec + 2, E.toz);
                            agent_storef(rec + 3, __uint_as_float(gen + 1u));
                            agent_storef(rec + 4, E.tdx); agent_storef(rec + 5, E.tdy); agent_storef(rec + 6, E.tdz);
                            agent_storef(rec + 7, __uint_as_float(nb + 1u));
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
            if (go != 1u) { overflowed = go == 2u; break; }
            R.ox = uniform_f(nox); R.oy = uniform_f(noy); R.oz = uniform_f(noz);
            R.dx = uniform_f(ndx); R.dy = uniform_f(ndy); R.dz = uniform_f(ndz);
            node = __builtin_amdgcn_readfirstlane(next_node);
            gen = gen + 1u;
            chained++;
        }
        finished++;
        if (overflowed) break;
    }
    if (lane == 0) {
        if (finished) atomicAdd(done, finished);
        if (chained) atomicAdd(A.s_counts + SC_DEEP_CHAINED, chained);  // diagnostics: rays that never saw the queue,
        if (waits) atomicAdd(A.s_counts + SC_DEEP_WAITS, waits);        // claims that had to wait for their record
    }
}
