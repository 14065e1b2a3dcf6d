// kernel_lane.h -- render_lanes: one independent ray per lane.
//
// This kernel can render any pixel (it is the complete hot path, queue-less): the fallback of the ray-stream pass after a
// queue overflow, the GI mode, and crt_tuning::mode = lanes.  A lane owns a pixel until
// its colour is final, walking the reference's recursion with an explicit frame stack
// (RayTracer.cpp:358-451); a lane that finishes fetches the next pixel from a global counter, so a
// wavefront stays full until the work runs out.
#pragma once

#include "kernel_common.h"
#include "kernel_walk.h"
#include "gi_random.h"
#include "glibc_sincosf.h"

// GI frames (FR_DIFFUSE_GI and, in the GI mode, every other kind) are FRAME_DWORDS_GI wide:
//   [0] kind   [1..3] the direct light sum (finalColor)   [4..6] the indirect sum   [7] samples done
//   [8] the invocation's key (gi_random.h; every kind keeps it here)   [9..11] hit point   [12..14] hit normal   [15..17] incoming direction
enum : int { FR_DIFFUSE_GI = 4 };

// RayTracer::getRay with the pixel offsets given (RayTracer.cpp:61-80; 0.5, 0.5 = primary_ray), then shootRay's own normalisation
__device__ __forceinline__ void primary_ray_offset(const KernelArgs &A, uint32_t px, uint32_t py, float offx, float offy, Ray &R) {
    float x = (float)px + offx;
    float y = (float)py + offy;
    x = x / (float)A.s->width;
    y = y / (float)A.s->height;
    x = (2.0f * x) - 1.0f;
    y = 1.0f - (2.0f * y);
    x = x * ((float)A.s->width / (float)A.s->height);
    const float z = -1.0f;
    R.dx = x * A.f->cam[0] + y * A.f->cam[3] + z * A.f->cam[6];
    R.dy = x * A.f->cam[1] + y * A.f->cam[4] + z * A.f->cam[7];
    R.dz = x * A.f->cam[2] + y * A.f->cam[5] + z * A.f->cam[8];
    normalize3(R.dx, R.dy, R.dz);
    R.ox = A.f->cam_pos[0]; R.oy = A.f->cam_pos[1]; R.oz = A.f->cam_pos[2];
    normalize3(R.dx, R.dy, R.dz);
    ray_prepare(R);
}

// The direction of GI sample i at a diffuse hit (RayTracer.cpp:334-347): a vector in the XY half-plane turned around Y, taken
// from the hit's local frame {right = normalize(d x n), up = n, forward = right x up} to world space.  (d: the incoming ray's
// direction, n: the hit normal, u1 / u2: the sample's two uniform numbers.)
__device__ __forceinline__ void gi_sample_direction(float dx, float dy, float dz, float nx, float ny, float nz, float u1, float u2,
                                                    float &ox, float &oy, float &oz) {
    float rx = dy * nz - dz * ny, ry = dz * nx - dx * nz, rz = dx * ny - dy * nx;  // Vector::operator*(Vector), Vector.cpp:61-65
    normalize3(rx, ry, rz);
    const float fx = ry * nz - rz * ny, fy = rz * nx - rx * nz, fz = rx * ny - ry * nx;
    const float angle1 = PI_F * u1;
    const float angle2 = 2 * PI_F * u2;
    const float vx = crt_cosf(angle1), vy = crt_sinf(angle1), vz = 0.0f;
    const float c2 = crt_cosf(angle2), s2 = crt_sinf(angle2);
    // randomVectorInXY * rotateAroundY, rows {c2, 0, -s2} {0, 1, 0} {s2, 0, c2} (Matrix.h:137-142)
    const float qx = vx * c2 + vy * 0.0f + vz * s2;
    const float qy = vx * 0.0f + vy * 1.0f + vz * 0.0f;
    const float qz = vx * -s2 + vy * 0.0f + vz * c2;
    // ... * localHitMatrix, rows right / up / forward
    ox = qx * rx + qy * nx + qz * fx;
    oy = qx * ry + qy * ny + qz * fy;
    oz = qx * rz + qy * nz + qz * fz;
}

struct LaneWalk {
    // traversal cursors: top-level node / position in a top-level leaf / mesh-tree node / position in a mesh leaf
    uint32_t tnode, tleaf, mnode, mleaf, cur_mesh;
    // mesh-level and scene-level running closest hit (KDTree.cpp:75-86, 156-167)
    bool mhave, have, occluded;
    float mmin, mt, tmin, bt, light_dist;
    uint32_t mtri, btri, bmesh;
    int rtype;
    SeenMeshes seen;   // meshes already walked for this ray (kernel_common.h: mesh_walk_is_repeat; production build only)
};

__device__ __forceinline__ void traversal_begin(LaneWalk &L, uint32_t top_root) {
    L.tnode = top_root;
    L.tleaf = NONE;
    L.mnode = END;
    L.mleaf = NONE;
    L.cur_mesh = NONE;
    L.mhave = false;
    L.have = false;
    L.occluded = false;
    L.tmin = INFINITY;
    L.mmin = INFINITY;
    seen_clear(L.seen);
}

// One unit of traversal work for this lane: either one triangle test (when inside a leaf) or one
// node visit / bookkeeping step.  Returns false when the whole two-level walk is finished.
template <bool COUNT>
__device__ __forceinline__ bool traversal_step(LaneWalk &L, const Ray &R, const KernelArgs &A, uint32_t *cnt) {
    if (L.mleaf != NONE) {
        // ---- inside a mesh-tree leaf: test one triangle (KDTree.cpp:57-65)
        // the leaf's triangles are stored in list order, one 64-byte record each: no index indirection
        const float4 a = A.s->ltris[4 * (size_t)L.mleaf + 0];
        const float4 b = A.s->ltris[4 * (size_t)L.mleaf + 1];
        const float4 c = A.s->ltris[4 * (size_t)L.mleaf + 2];
        const float4 d = A.s->ltris[4 * (size_t)L.mleaf + 3];
        const float plane = d.x;
        const uint32_t tri = __float_as_uint(d.y);
        L.mleaf = __float_as_uint(d.z) ? NONE : L.mleaf + 1;
        if (COUNT) { cnt[C_TRI]++; cnt[C_LEAFIDX]++; }
        float t;
        if (triangle_test(R, L.rtype == RAY_PRIMARY, a, b, c, plane, t)) {
            // `closest = hits[0]; min = inf; for h: if (h.d < min) {min = h.d; closest = h}` fused into the walk
            if (!L.mhave) { L.mhave = true; L.mt = t; L.mtri = tri; }
            if (t < L.mmin) { L.mmin = t; L.mt = t; L.mtri = tri; }
            // production build: a shadow walk ends at the first accepted hit within the light's distance -- exact, by the
            // monotonicity argument at kernel_walk.h: shadow_hit_occludes (the counting build walks on, as the reference does)
            if (!COUNT && L.rtype == RAY_SHADOW && t < INFINITY &&
                shadow_hit_occludes(R, R.ox + R.dx * t, R.oy + R.dy * t, R.oz + R.dz * t, L.light_dist)) {
                L.occluded = true;
                L.mleaf = NONE; L.cur_mesh = NONE; L.mnode = END; L.tleaf = NONE; L.tnode = END;  // the next step reports the end
            }
        }
        return true;
    }
    if (L.cur_mesh != NONE) {
        if (L.mnode != END) {
            // ---- visit one mesh-tree node (KDTree.cpp:53-74)
            const float4 q0 = A.s->nodes[2 * (size_t)L.mnode], q1 = A.s->nodes[2 * (size_t)L.mnode + 1];
            const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
            if (COUNT) cnt[C_BOX]++;
            const bool hit = slab_test(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
            if (hit && is_leaf_link(link)) {
                L.mleaf = link & ~LEAF;
                L.mnode = miss;
            } else {
                L.mnode = hit ? link : miss;
            }
            return true;
        }
        // ---- this mesh is finished: hand its closest hit to the scene level
        if (L.mhave) {
            if (L.rtype == RAY_SHADOW) {
                // AccelerationStructure.cpp:73-74: length(hitPoint - origin) <= distanceToLight
                const float px = R.ox + R.dx * L.mt, py = R.oy + R.dy * L.mt, pz = R.oz + R.dz * L.mt;
                if (len3(px - R.ox, py - R.oy, pz - R.oz) <= L.light_dist) L.occluded = true;
            } else {
                if (!L.have) { L.have = true; L.bt = L.mt; L.btri = L.mtri; L.bmesh = L.cur_mesh; }
                if (L.mt < L.tmin) { L.tmin = L.mt; L.bt = L.mt; L.btri = L.mtri; L.bmesh = L.cur_mesh; }
            }
        }
        L.cur_mesh = NONE;
        return true;
    }
    if (L.tleaf != NONE) {
        // ---- inside a top-level leaf: start the next mesh (KDTree.cpp:138-144, AccelerationStructure.cpp:66-72)
        const uint32_t ent = A.s->leaf_meshes[L.tleaf];
        const uint32_t mi = ent & ~LAST;
        L.tleaf = (ent & LAST) ? NONE : L.tleaf + 1;
        if (COUNT) cnt[C_LEAFIDX]++;
        const crt_mesh m = A.s->meshes[mi];
        if (L.rtype == RAY_SHADOW && (m.flags & 1u) && !A.f->use_gi) return true;  // AccelerationStructure.cpp:67-71
        if (!COUNT && mesh_walk_is_repeat(L.seen, mi)) return true;  // production build: every mesh once per ray (kernel_common.h, exact)
        L.cur_mesh = mi;
        L.mnode = m.root;
        L.mhave = false;
        L.mmin = INFINITY;
        return true;
    }
    if (L.tnode != END) {
        // ---- visit one top-level node (KDTree.cpp:132-155)
        const float4 q0 = A.s->nodes[2 * (size_t)L.tnode], q1 = A.s->nodes[2 * (size_t)L.tnode + 1];
        const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
        if (COUNT) cnt[C_BOX]++;
        const bool hit = slab_test(R, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z);
        if (hit && is_leaf_link(link)) {
            L.tleaf = link & ~LEAF;
            L.tnode = miss;
        } else {
            L.tnode = hit ? link : miss;
        }
        return true;
    }
    return false;
}

// GI: the reference's GI / multi-sample mode (RayTracer.cpp:90-104, 331-354) with the counter-based generator of gi_random.h
template <bool COUNT, bool GI = false>
__global__ __launch_bounds__(BLOCK) void render_lanes(const KernelArgs A) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    float *frames = A.f->frames + (size_t)wave * A.f->frame_wave_stride + lane;
    constexpr int FDW = GI ? FRAME_DWORDS_GI : FRAME_DWORDS;
    auto FR = [&](uint32_t level, int field) -> float & { return frames[((size_t)level * FDW + field) * 64]; };
    auto FRK = [&](uint32_t level) -> int & { return *reinterpret_cast<int *>(&frames[(size_t)level * FDW * 64]); };
    auto FRU = [&](uint32_t level, int field) -> uint32_t & { return *reinterpret_cast<uint32_t *>(&frames[((size_t)level * FDW + field) * 64]); };
    // GI state: the current invocation's key; the pixel being sampled
    uint32_t key = 0, pixel_key = 0, sample = 0, cur_px = 0, cur_py = 0;
    float sumx = 0, sumy = 0, sumz = 0;          // std::accumulate over the pixel's samples (RayTracer.cpp:102-104)
    float inx = 0, iny = 0, inz = 0;             // the ray direction that reached the diffuse hit being lit
    const uint32_t n_samples = A.f->rays_per_pixel ? A.f->rays_per_pixel : 1u;  // colorVector always holds the centre sample

    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;

    Ray R;
    LaneWalk L;
    int state = ST_FETCH;
    uint32_t sp = 0;           // recursion level of the current ray == number of frames below it
    size_t out_off = 0;
    // diffuse light loop (RayTracer.cpp:300-330)
    float hpx = 0, hpy = 0, hpz = 0, hnx = 0, hny = 0, hnz = 0;
    float basex = 0, basey = 0, basez = 0, accx = 0, accy = 0, accz = 0, kfac = 0;
    uint32_t li = 0;
    bool base_is_bitmap = false;
    if (A.only_if_overflow) {  // fallback of the stream pass: usually nothing to do
        if (!A.f->s_counts[SC_OVERFLOW_WORD]) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(A.f->fallback_total, 1u);  // reported as crt_stats::fallback_frames
    }
    const uint32_t total_px = A.f->n_items * 64u;

    for (;;) {
        // ------------------------------------------------------------------ fetch new pixels
        if (__ballot(state == ST_FETCH)) {
            while (state == ST_FETCH) {
                const unsigned long long need = __ballot(1);
                const int n = __popcll(need);
                const int rank = __popcll(need & ((1ull << lane) - 1ull));
                uint32_t base = 0;
                if (rank == 0) base = atomicAdd(A.f->pixel_counter, (uint32_t)n);
                base = __shfl(base, __ffsll((long long)need) - 1);
                uint32_t q = base + (uint32_t)rank;
                if (q >= total_px) { state = ST_DONE; break; }
                const WorkItem wi = A.f->items[q >> 6];
                const uint32_t sub = q & 63u;
                const uint32_t px = (wi.tile % A.s->tiles_x) * TILE + (sub & 7u);
                const uint32_t py = (wi.tile / A.s->tiles_x) * TILE + (sub >> 3);
                if (!((wi.mask >> sub) & 1ull) || px >= A.s->width || py >= A.s->height) continue;  // not covered: take another
                out_off = A.f->packed ? ((size_t)wi.out_tile * 64 + sub) * 3 : ((size_t)py * A.s->width + px) * 3;
                primary_ray(A, px, py, R);  // depth 0 <= MAX_DEPTH always (RayTracer.cpp:427)
                if (GI) {
                    cur_px = px; cur_py = py;
                    pixel_key = crt_gi_mix(A.f->gi_seed, py * A.s->width + px);
                    sample = 0;
                    key = crt_gi_mix(pixel_key, 0u);
                    sumx = sumy = sumz = 0;
                }
                L.rtype = RAY_PRIMARY;
                sp = 0;
                if (COUNT) cnt[C_PRIMARY]++;
                traversal_begin(L, A.s->top_root);
                state = ST_TRAVERSE;
            }
        }
        if (!__ballot(state != ST_DONE)) break;

        // ------------------------------------------------------------------ traverse
        bool finished = false;
        if (state == ST_TRAVERSE) {
            int budget = 48;
            do {
                if (!traversal_step<COUNT>(L, R, A, cnt)) { finished = true; break; }
            } while (--budget > 0);
        }

        // ------------------------------------------------------------------ a walk ended: shade / continue
        if (finished) {
            bool returning = false;      // a colour is being returned to the caller level
            bool new_ray = false;        // R holds a new ray that enters shootRay at level sp
            bool next_light = false;
            float cx = 0, cy = 0, cz = 0;

            if (L.rtype == RAY_SHADOW) {
                if (!L.occluded) {  // RayTracer.cpp:319-328
                    if (COUNT && base_is_bitmap) cnt[C_TEXEL]++;
                    accx += kfac * basex; accy += kfac * basey; accz += kfac * basez;
                }
                li++;
                next_light = true;
            } else if (!L.have) {
                cx = A.s->bgx; cy = A.s->bgy; cz = A.s->bgz; returning = true;  // RayTracer.cpp:449-450
            } else {
                Surface S;
                surface_at(A, R, L.bt, L.btri, L.bmesh, S);
                if (COUNT) cnt[C_HIT]++;
                if (S.M.type == CRT_MAT_DIFFUSE) {
                    hpx = S.px; hpy = S.py; hpz = S.pz; hnx = S.nx; hny = S.ny; hnz = S.nz;
                    if (GI) { inx = R.dx; iny = R.dy; inz = R.dz; }
                    base_is_bitmap = false;
                    if (S.M.texture >= 0) {
                        texture_color<COUNT>(A, A.s->textures[S.M.texture], L.btri, S.u, S.v, 1.0f - S.u - S.v, basex, basey,
                                             basez, base_is_bitmap);
                    } else { basex = S.M.ax; basey = S.M.ay; basez = S.M.az; }
                    accx = accy = accz = 0;
                    li = 0;
                    next_light = true;
                } else if (S.M.type == CRT_MAT_REFLECTIVE) {
                    // RayTracer::calculateReflection (RayTracer.cpp:358-374)
                    FRK(sp) = FR_REFLECT;
                    FR(sp, 1) = S.M.ax; FR(sp, 2) = S.M.ay; FR(sp, 3) = S.M.az;
                    if (GI) { FRU(sp, 8) = key; key = crt_gi_child_key(key, 0u); }
                    const float k = 2 * dot3(R.dx, R.dy, R.dz, S.nx, S.ny, S.nz);  // Vector::reflect, Vector.cpp:119-122
                    const float rx = R.dx - k * S.nx, ry = R.dy - k * S.ny, rz = R.dz - k * S.nz;
                    R.ox = S.px + S.nx * A.f->reflection_bias; R.oy = S.py + S.ny * A.f->reflection_bias; R.oz = S.pz + S.nz * A.f->reflection_bias;
                    R.dx = rx; R.dy = ry; R.dz = rz;
                    normalize3(R.dx, R.dy, R.dz);
                    L.rtype = RAY_REFLECTION;
                    sp++;
                    new_ray = true;
                } else if (S.M.type == CRT_MAT_REFRACTIVE) {
                    // RayTracer::calculateRefraction (RayTracer.cpp:375-417)
                    float nx = S.nx, ny = S.ny, nz = S.nz;
                    float eta1 = 1.0f, eta2 = S.M.ior;
                    float idn = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
                    if (idn > 0) {
                        const float s = eta1; eta1 = eta2; eta2 = s;
                        nx = -1.0f * nx; ny = -1.0f * ny; nz = -1.0f * nz;
                        idn = -idn;
                    }
                    const float cos_a = -idn;
                    const float sin_a = sqrtf(std_max(0.0f, 1 - cos_a * cos_a));
                    const float k = 2 * dot3(R.dx, R.dy, R.dz, nx, ny, nz);
                    const float rx = R.dx - k * nx, ry = R.dy - k * ny, rz = R.dz - k * nz;
                    const float eta_ratio = eta1 / eta2;
                    const float sin_b = eta_ratio * sin_a;
                    if (sin_b < 1.0f) {
                        const float q = (eta1 - eta2) / (eta1 + eta2);
                        const float r0 = q * q;  // std::powf(q, 2): folded to q*q by the reference's compiler at -O2
                        const float fresnel = r0 + (1 - r0) * crt_pow5(1.0f - cos_a);
                        const float cos_b = sqrtf(std_max(0.0f, 1 - sin_b * sin_b));
                        float tx = eta_ratio * (R.dx + cos_a * nx) - cos_b * nx;
                        float ty = eta_ratio * (R.dy + cos_a * ny) - cos_b * ny;
                        float tz = eta_ratio * (R.dz + cos_a * nz) - cos_b * nz;
                        normalize3(tx, ty, tz);
                        FRK(sp) = FR_REFRACT_WAIT_REFLECTION;
                        FR(sp, 1) = S.px - nx * A.f->refraction_bias; FR(sp, 2) = S.py - ny * A.f->refraction_bias;
                        FR(sp, 3) = S.pz - nz * A.f->refraction_bias;
                        FR(sp, 4) = tx; FR(sp, 5) = ty; FR(sp, 6) = tz;
                        FR(sp, 7) = fresnel;
                    } else {
                        FRK(sp) = FR_REFRACT_NO_TRANSMISSION;
                    }
                    R.ox = S.px + nx * A.f->reflection_bias; R.oy = S.py + ny * A.f->reflection_bias; R.oz = S.pz + nz * A.f->reflection_bias;
                    R.dx = rx; R.dy = ry; R.dz = rz;
                    normalize3(R.dx, R.dy, R.dz);
                    L.rtype = RAY_REFLECTION;
                    if (GI) { FRU(sp, 8) = key; key = crt_gi_child_key(key, 0u); }
                    sp++;
                    new_ray = true;
                } else {
                    cx = A.s->bgx; cy = A.s->bgy; cz = A.s->bgz; returning = true;  // RayTracer.cpp:443-446
                }
            }

            // ---- diffuse light loop: set up the next shadow ray or return the accumulated colour
            if (next_light) {
                if (li < A.s->n_lights) {
                    if (COUNT) { cnt[C_LIGHT]++; cnt[C_SHADOW]++; }
                    light_setup(A, li, hpx, hpy, hpz, hnx, hny, hnz, R, L.light_dist, kfac);
                    L.rtype = RAY_SHADOW;
                    traversal_begin(L, A.s->top_root);
                } else if (GI && A.f->use_gi) {
                    // RayTracer.cpp:331-354: GI_SAMPLE_SIZE diffuse reflection rays, one after the other, each a shootRay(depth + 1)
                    if (A.f->gi_samples == 0) {
                        const float inv = 1.0f / (float)(A.f->gi_samples + 1u);
                        cx = (accx + 0.0f) * inv; cy = (accy + 0.0f) * inv; cz = (accz + 0.0f) * inv;
                        returning = true;
                    } else {
                        FRK(sp) = FR_DIFFUSE_GI;
                        FR(sp, 1) = accx; FR(sp, 2) = accy; FR(sp, 3) = accz;
                        FR(sp, 4) = 0.0f; FR(sp, 5) = 0.0f; FR(sp, 6) = 0.0f;
                        FRU(sp, 7) = 0u;
                        FRU(sp, 8) = key;
                        FR(sp, 9) = hpx; FR(sp, 10) = hpy; FR(sp, 11) = hpz;
                        FR(sp, 12) = hnx; FR(sp, 13) = hny; FR(sp, 14) = hnz;
                        FR(sp, 15) = inx; FR(sp, 16) = iny; FR(sp, 17) = inz;
                        gi_sample_direction(inx, iny, inz, hnx, hny, hnz, crt_gi_uniform(key, 2u), crt_gi_uniform(key, 3u), R.dx, R.dy, R.dz);
                        R.ox = hpx + hnx * A.f->monte_carlo_bias; R.oy = hpy + hny * A.f->monte_carlo_bias; R.oz = hpz + hnz * A.f->monte_carlo_bias;
                        L.rtype = RAY_REFLECTION;
                        key = crt_gi_child_key(key, 2u);
                        sp++;
                        new_ray = true;
                    }
                } else {
                    cx = accx; cy = accy; cz = accz; returning = true;
                }
            }

            // ---- unwind / advance the explicit recursion (post-order, as the reference's call stack does)
            while (returning || new_ray) {
                if (new_ray) {
                    // shootRay entry (RayTracer.cpp:419-429)
                    normalize3(R.dx, R.dy, R.dz);
                    new_ray = false;
                    if (sp > A.f->max_depth) { cx = A.s->bgx; cy = A.s->bgy; cz = A.s->bgz; returning = true; continue; }
                    if (COUNT) cnt[C_SECONDARY]++;
                    ray_prepare(R);
                    traversal_begin(L, A.s->top_root);
                    break;
                }
                if (sp == 0) {
                    if (GI && A.f->use_gi) {  // RayTracer.cpp:90-104: the centre sample, then RAYS_PER_PIXEL - 1 jittered ones, then the mean
                        sumx = sumx + cx; sumy = sumy + cy; sumz = sumz + cz;
                        sample++;
                        if (sample < n_samples) {
                            key = crt_gi_mix(pixel_key, sample);
                            primary_ray_offset(A, cur_px, cur_py, crt_gi_uniform(key, 0u), crt_gi_uniform(key, 1u), R);
                            L.rtype = RAY_PRIMARY;
                            if (COUNT) cnt[C_PRIMARY]++;
                            traversal_begin(L, A.s->top_root);
                            returning = false;
                            break;
                        }
                        const float inv = 1.0f / (float)n_samples;
                        cx = sumx * inv; cy = sumy * inv; cz = sumz * inv;
                    }
                    A.f->out[out_off] = cx; A.f->out[out_off + 1] = cy; A.f->out[out_off + 2] = cz;  // RayTracer.cpp:106
                    state = ST_FETCH;
                    break;
                }
                const uint32_t f = sp - 1;
                const int kind = FRK(f);
                if (GI) key = FRU(f, 8);  // back in the caller's invocation
                if (GI && kind == FR_DIFFUSE_GI) {
                    const float ix = FR(f, 4) + cx, iy = FR(f, 5) + cy, iz = FR(f, 6) + cz;  // indirectLightContribution += shootRay(...)
                    const uint32_t i = FRU(f, 7) + 1u;
                    if (i < A.f->gi_samples) {
                        FR(f, 4) = ix; FR(f, 5) = iy; FR(f, 6) = iz;
                        FRU(f, 7) = i;
                        const float px = FR(f, 9), py = FR(f, 10), pz = FR(f, 11), nx = FR(f, 12), ny = FR(f, 13), nz = FR(f, 14);
                        gi_sample_direction(FR(f, 15), FR(f, 16), FR(f, 17), nx, ny, nz, crt_gi_uniform(key, 2u + 2u * i),
                                            crt_gi_uniform(key, 3u + 2u * i), R.dx, R.dy, R.dz);
                        R.ox = px + nx * A.f->monte_carlo_bias; R.oy = py + ny * A.f->monte_carlo_bias; R.oz = pz + nz * A.f->monte_carlo_bias;
                        L.rtype = RAY_REFLECTION;
                        key = crt_gi_child_key(key, 2u + i);
                        returning = false;
                        new_ray = true;   // enters shootRay at level sp (== f + 1)
                    } else {
                        const float inv = 1.0f / (float)(A.f->gi_samples + 1u);  // RayTracer.cpp:352-353
                        cx = (FR(f, 1) + ix) * inv; cy = (FR(f, 2) + iy) * inv; cz = (FR(f, 3) + iz) * inv;
                        sp = f;
                    }
                } else if (kind == FR_REFLECT) {
                    cx = 0.0f + FR(f, 1) * cx; cy = 0.0f + FR(f, 2) * cy; cz = 0.0f + FR(f, 3) * cz;  // RayTracer.cpp:368-372
                    sp = f;
                } else if (kind == FR_REFRACT_NO_TRANSMISSION) {
                    sp = f;  // `return reflectionColor`, RayTracer.cpp:416
                } else if (kind == FR_REFRACT_WAIT_REFLECTION) {
                    R.ox = FR(f, 1); R.oy = FR(f, 2); R.oz = FR(f, 3);
                    R.dx = FR(f, 4); R.dy = FR(f, 5); R.dz = FR(f, 6);
                    L.rtype = RAY_REFRACTION;
                    if (GI) key = crt_gi_child_key(key, 1u);
                    FRK(f) = FR_REFRACT_WAIT_REFRACTION;
                    FR(f, 1) = cx; FR(f, 2) = cy; FR(f, 3) = cz;  // reflectionColor
                    returning = false;
                    new_ray = true;   // enters shootRay at level sp (== f + 1)
                } else {
                    const float fr = FR(f, 7);  // RayTracer.cpp:414
                    cx = fr * FR(f, 1) + (1 - fr) * cx; cy = fr * FR(f, 2) + (1 - fr) * cy; cz = fr * FR(f, 3) + (1 - fr) * cz;
                    sp = f;
                }
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
