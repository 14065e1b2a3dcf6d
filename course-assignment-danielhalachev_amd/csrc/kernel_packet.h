// kernel_packet.h -- render_packets: one wavefront per 8x8 pixel tile, 64 coherent rays walked together.
//
// The reference's tree walk has a fixed visit order and never prunes by distance (KDTree.cpp:53-74), so
// all 64 rays of a tile can share ONE walk: the wave visits a node if any of its rays passes the box
// test of all ancestors, and a ray that failed a box test simply sits out until the walk leaves that
// subtree.  Because nodes are laid out in visit order, "leaves that subtree" is `node index >= miss
// link of the failed node`: one integer per lane replaces the per-ray stack.
//
// What this buys on CDNA4: the node and triangle records are wave-uniform, so they are fetched with
// scalar loads (s_load_dwordx8/x16 through the scalar cache) into SGPRs -- no per-lane gathers, no
// VGPRs for geometry -- and every VALU instruction tests one box or one triangle against 64 rays.
// Primary rays and the shadow rays of a tile's diffuse hits (85-90 % of all rays of the benchmark
// frames) are coherent enough for this; pixels whose primary hit is reflective or refractive are
// handed to render_lanes (kernel_lane.h), which re-renders them from the primary ray.
#pragma once

#include "kernel_common.h"

typedef const float __attribute__((address_space(4))) *kfp;      // constant address space: loads with a
typedef const uint32_t __attribute__((address_space(4))) *ku32p;  // wave-uniform address become scalar loads
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const v8f __attribute__((address_space(4))) *kv8p;
typedef const v16f __attribute__((address_space(4))) *kv16p;

// Whole records in ONE scalar load each (s_load_dwordx8 / s_load_dwordx16): loading the fields one by one
// lets the compiler sink each dword load into the branch that first uses it, one wait per field.
__device__ __forceinline__ v8f load_node(kfp nodes, uint32_t index) { return *(kv8p)(nodes + 8 * (size_t)index); }
__device__ __forceinline__ v16f load_triangle(kfp tris, uint32_t index) { return *(kv16p)(tris + 16 * (size_t)index); }

// Walks the two-level tree once for the 64 rays of the wave.  `on`: this lane has a ray.
// SHADOW = false: closest hit (KDTree.cpp:127-167) -> have / bt / btri / bmesh.
// SHADOW = true : ObjectKDTree::checkForIntersection (AccelerationStructure.cpp:56-94) -> occluded.
// TRACK: keep, per lane, the node index from which it takes part again after a failed box test.  In a
//   tree whose child boxes lie inside their parent's box (every tree the reference builds: a child box is
//   its parent's with one face moved inwards, BoundingBox.h:60-69) this is redundant: the slab test is
//   monotone under rounding, so a ray that fails a box fails every box nested in it, and the plain test
//   `on && slab(node)` already is "passed all ancestors".  crt_create checks the nesting; TRACK is used
//   when it does not hold, and in the counting build (to count a box test only where the reference runs one).
// PAR: some lane has a direction component below FLT_EPSILON (BoundingBox.h:90) -> general slab test.
template <bool SHADOW, bool PRIMARY, bool COUNT, bool TRACK, bool PAR>
__device__ __forceinline__ void packet_walk_impl(const KernelArgs &A, const Ray &R, const bool on, const float light_dist,
                                                 bool &have, float &bt, uint32_t &btri, uint32_t &bmesh, bool &occluded,
                                                 uint32_t *cnt, int &budget) {
    const kfp nodes = (kfp)(const float *)A.nodes;
    const kfp ltris = (kfp)(const float *)A.ltris;
    const ku32p leaf_meshes = (ku32p)A.leaf_meshes;
    const ku32p meshes = (ku32p)(const uint32_t *)A.meshes;  // crt_mesh = 4 x u32: root, material, flags, pad
    auto slab = [&](const v8f &q) -> bool {
        return PAR ? slab_test_general(R, q[0], q[1], q[2], q[4], q[5], q[6])
                   : slab_test_no_parallel(R, q[0], q[1], q[2], q[4], q[5], q[6]);
    };

    have = false;
    occluded = false;
    if (COUNT && (threadIdx.x & 63u) == 0) cnt[C_WAVE_WALKS]++;
    float tmin = INFINITY;
    SeenMeshes seen;                // per lane: meshes already walked for this ray (kernel_common.h: mesh_walk_is_repeat)
    seen_clear(seen);
    uint32_t tres = on ? 0u : END;  // TRACK: the lane takes part in the top-level walk from node index `tres` on
    uint32_t ti = A.top_root;
    while (ti != END && budget >= 0) {
        const v8f q = load_node(nodes, ti);
        const uint32_t miss = __float_as_uint(q[3]), link = __float_as_uint(q[7]);
        if (COUNT && (threadIdx.x & 63u) == 0) cnt[C_WAVE_NODES]++;
        // (production build: a shadow ray that is already occluded has its answer -- an OR over the meshes -- and sits out)
        const bool act = (TRACK ? (ti >= tres) : on) && !(SHADOW && !COUNT && occluded);
        const bool hit = act && slab(q);
        if (COUNT && act) cnt[C_BOX]++;
        if (TRACK && act && !hit) tres = miss;
        if (!__ballot(hit)) { ti = miss; continue; }
        if (!is_leaf_link(link)) { ti = link; continue; }
        uint32_t e = link & ~LEAF;
        for (;;) {  // meshes of this top-level leaf, in list order (KDTree.cpp:138-144)
            const uint32_t ent = leaf_meshes[e++];
            const uint32_t mi = ent & ~LAST;
            const uint32_t mroot = meshes[4 * (size_t)mi], mflags = meshes[4 * (size_t)mi + 2];
            if (COUNT && hit) cnt[C_LEAFIDX]++;
            bool mon = hit && !(SHADOW && (mflags & 1u));  // shadow rays skip refractive meshes
            if (!COUNT && mon && mesh_walk_is_repeat(seen, mi)) mon = false;  // production build: every mesh once per ray
            if (__ballot(mon)) {
                // ---- one mesh tree (KDTree.cpp:48-87), closest hit per lane with the reference's tie rule
                bool mhave = false;
                float mmin = INFINITY, mt = 0;
                uint32_t mtri = 0;
                uint32_t mres = mon ? mroot : END;
                uint32_t i = mroot;
                while (i != END && budget >= 0) {
                    budget--;
                    const v8f n = load_node(nodes, i);
                    const uint32_t nmiss = __float_as_uint(n[3]), nlink = __float_as_uint(n[7]);
                    if (COUNT && (threadIdx.x & 63u) == 0) cnt[C_WAVE_NODES]++;
                    const bool nact = TRACK ? (i >= mres) : mon;
                    const bool nhit = nact && slab(n);
                    if (COUNT && nact) cnt[C_BOX]++;
                    if (TRACK && nact && !nhit) mres = nmiss;
                    if (!__ballot(nhit)) { i = nmiss; continue; }
                    if (!is_leaf_link(nlink)) { i = nlink; continue; }
                    uint32_t te = nlink & ~LEAF;
                    for (;;) {  // triangles of this leaf, in list order (KDTree.cpp:57-65)
                        budget--;
                        const v16f T = load_triangle(ltris, te++);
                        const uint32_t tri = __float_as_uint(T[13]);
                        const uint32_t tent = __float_as_uint(T[14]) ? LAST : 0u;
                        const float4 a = make_float4(T[0], T[1], T[2], T[3]), b = make_float4(T[4], T[5], T[6], T[7]),
                                     c = make_float4(T[8], T[9], T[10], T[11]);
                        const float plane = T[12];
                        if (COUNT && (threadIdx.x & 63u) == 0) cnt[C_WAVE_TRIS]++;
                        if (nhit) {
                            if (COUNT) { cnt[C_TRI]++; cnt[C_LEAFIDX]++; }
                            float t;
                            if (triangle_test(R, PRIMARY, a, b, c, plane, t)) {
                                if (!mhave) { mhave = true; mt = t; mtri = tri; }
                                if (t < mmin) { mmin = t; mt = t; mtri = tri; }
                            }
                        }
                        if (tent & LAST) break;
                    }
                    i = nmiss;
                }
                if (mon && mhave) {
                    if (SHADOW) {  // AccelerationStructure.cpp:73-74
                        const float px = R.ox + R.dx * mt, py = R.oy + R.dy * mt, pz = R.oz + R.dz * mt;
                        if (len3(px - R.ox, py - R.oy, pz - R.oz) <= light_dist) occluded = true;
                    } else {       // KDTree.cpp:156-167
                        if (!have) { have = true; bt = mt; btri = mtri; bmesh = mi; }
                        if (mt < tmin) { tmin = mt; bt = mt; btri = mtri; bmesh = mi; }
                    }
                }
            }
            if (ent & LAST) break;
        }
        ti = miss;
    }
}

template <bool SHADOW, bool PRIMARY, bool COUNT>
__device__ __forceinline__ void packet_walk(const KernelArgs &A, const Ray &R, const bool on, const float light_dist,
                                            bool &have, float &bt, uint32_t &btri, uint32_t &bmesh, bool &occluded,
                                            uint32_t *cnt, int &budget) {
    const bool par = __ballot(on && R.parmask != 0) != 0;  // wave-uniform
    if (COUNT || !A.nested_boxes) {
        if (par) packet_walk_impl<SHADOW, PRIMARY, COUNT, true, true>(A, R, on, light_dist, have, bt, btri, bmesh, occluded, cnt, budget);
        else packet_walk_impl<SHADOW, PRIMARY, COUNT, true, false>(A, R, on, light_dist, have, bt, btri, bmesh, occluded, cnt, budget);
    } else {
        if (par) packet_walk_impl<SHADOW, PRIMARY, COUNT, false, true>(A, R, on, light_dist, have, bt, btri, bmesh, occluded, cnt, budget);
        else packet_walk_impl<SHADOW, PRIMARY, COUNT, false, false>(A, R, on, light_dist, have, bt, btri, bmesh, occluded, cnt, budget);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void render_packets(const KernelArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t cnt[C_N], saved[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;

    // static round-robin over the tiles (no atomic fetch + break on a wave-uniform value here: DESIGN.md "compiler notes")
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    for (uint32_t item = wave; item < A.n_items; item += n_waves) {
        const WorkItem wi = A.items[item];
        const uint32_t px = (wi.tile % A.tiles_x) * TILE + (lane & 7u);
        const uint32_t py = (wi.tile / A.tiles_x) * TILE + (lane >> 3);
        bool on = ((wi.mask >> lane) & 1ull) && px < A.width && py < A.height;
        const size_t out_off = A.packed ? ((size_t)wi.out_tile * 64 + lane) * 3 : ((size_t)py * A.width + px) * 3;

        // ---- primary rays of the tile (RayTracer.cpp:88-89)
        Ray R;
        primary_ray(A, px, py, R);
        if (COUNT) {
            for (int k = 0; k < C_N; k++) saved[k] = cnt[k];
            if (on) cnt[C_PRIMARY]++;
        }
        bool have, occluded;
        float bt = 0;
        uint32_t btri = 0, bmesh = 0;
        int budget = A.packet_budget ? (int)A.packet_budget : 0x7FFFFFFF;
        packet_walk<false, true, COUNT>(A, R, on, 0.0f, have, bt, btri, bmesh, occluded, cnt, budget);

        // ---- shootRay's dispatch on the material (RayTracer.cpp:431-450)
        float cx = A.bgx, cy = A.bgy, cz = A.bgz;
        bool diffuse = false, base_is_bitmap = false;
        float hpx = 0, hpy = 0, hpz = 0, hnx = 0, hny = 0, hnz = 0, basex = 0, basey = 0, basez = 0;
        if (on && have) {
            Surface S;
            surface_at(A, R, bt, btri, bmesh, S);
            if (S.M.type == CRT_MAT_REFLECTIVE || S.M.type == CRT_MAT_REFRACTIVE) {
                // incoherent from here on: hand the whole pixel to render_lanes, forget what was counted for it
                const uint32_t slot = atomicAdd(A.deferred_count, 1u);
                A.deferred[slot] = item * 64u + lane;
                on = false;
                if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = saved[k];
            } else {
                if (COUNT) cnt[C_HIT]++;
                if (S.M.type == CRT_MAT_DIFFUSE) {
                    diffuse = true;
                    hpx = S.px; hpy = S.py; hpz = S.pz; hnx = S.nx; hny = S.ny; hnz = S.nz;
                    if (S.M.texture >= 0) {
                        texture_color<COUNT>(A, A.textures[S.M.texture], btri, S.u, S.v, 1.0f - S.u - S.v, basex, basey,
                                             basez, base_is_bitmap);
                    } else { basex = S.M.ax; basey = S.M.ay; basez = S.M.az; }
                }
            }
        }

        // ---- calculateDiffusion: one shared walk per light for the tile's diffuse hits (RayTracer.cpp:308-330)
        if (__ballot(diffuse)) {
            float accx = 0, accy = 0, accz = 0;
            for (uint32_t li = 0; li < A.n_lights; li++) {
                Ray SR;
                float dist, kfac;
                light_setup(A, li, hpx, hpy, hpz, hnx, hny, hnz, SR, dist, kfac);
                if (COUNT && diffuse) { cnt[C_LIGHT]++; cnt[C_SHADOW]++; }
                bool shave, socc;
                float st;
                uint32_t stri, smesh;
                int sbudget = A.packet_budget ? (int)A.packet_budget : 0x7FFFFFFF;
                packet_walk<true, false, COUNT>(A, SR, diffuse, dist, shave, st, stri, smesh, socc, cnt, sbudget);
                if (diffuse && !socc) {
                    if (COUNT && base_is_bitmap) cnt[C_TEXEL]++;
                    accx += kfac * basex; accy += kfac * basey; accz += kfac * basez;
                }
            }
            if (diffuse) { cx = accx; cy = accy; cz = accz; }
        }
        if (on) { A.out[out_off] = cx; A.out[out_off + 1] = cy; A.out[out_off + 2] = cz; }  // RayTracer.cpp:106
    }

    if (COUNT) {
        for (int k = 0; k < C_N; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&A.counters[k], v);
        }
    }
}
