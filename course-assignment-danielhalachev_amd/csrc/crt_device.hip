// crt_device.hip -- the MI355X (gfx950) implementation behind include/crt_hip.h.
//
// What runs here is the reference's per-pixel hot path, tree mode, USE_GI = false
// (reference: SourceCode/src/RayTracer.cpp:61-112, 300-451, 507-517; src/KDTree.cpp:48-87,
// 127-192; src/AccelerationStructure.cpp:56-94; include/tracer/BoundingBox.h:85-108;
// src/Ray.cpp:9-31; src/Triangle.cpp:37-73; src/Texture.cpp:14-72; src/Color.cpp:12-16).
//
// Design (see DESIGN.md for the measurements behind it):
//  * one ray per lane, 64 independent rays per wavefront; a lane that finishes its pixel
//    fetches the next one from a global pixel counter, so waves stay full to the end of the frame;
//  * the reference's stack DFS has a FIXED visit order and no distance pruning, so the tree is
//    flattened into hit/miss links (crt_node) and walked without any stack;
//  * the binary reflect/refract recursion is an explicit per-lane frame stack, evaluated in the
//    reference's post-order so that every float is combined in the same order;
//  * arithmetic is IEEE binary32 with no contraction (-ffp-contract=off), correctly rounded
//    divide and sqrt, std::min/std::max semantics written out, so results are bit-identical to
//    the x86-64 reference build.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/crt_hip.h"
#include "glibc_powf.h"

namespace {

constexpr uint32_t END = CRT_LINK_END;
constexpr uint32_t LEAF = CRT_LINK_LEAF;
constexpr uint32_t LAST = CRT_ENTRY_LAST;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int TILE = 8;             // 8x8 pixel tiles: 64 pixels = one wavefront's worth
constexpr int FRAME_DWORDS = 8;     // per recursion level and lane
constexpr int BLOCK = 256;
constexpr float PI_F = 3.14159265358979323846f;  // M_PIf, RayTracer.cpp:27

enum : int { RAY_PRIMARY = 0, RAY_SHADOW = 1, RAY_REFLECTION = 2, RAY_REFRACTION = 3 };  // Ray.h:14
enum : int { ST_FETCH = 0, ST_TRAVERSE = 1, ST_DONE = 2 };
enum : int { FR_REFLECT = 0, FR_REFRACT_WAIT_REFLECTION = 1, FR_REFRACT_WAIT_REFRACTION = 2, FR_REFRACT_NO_TRANSMISSION = 3 };
enum : int { C_BOX = 0, C_TRI, C_LEAFIDX, C_HIT, C_LIGHT, C_TEXEL, C_PRIMARY, C_SECONDARY, C_SHADOW, C_N };

struct DMaterial { float ax, ay, az, ior; uint32_t type, smooth; int32_t texture; uint32_t pad; };
struct DTexture { uint32_t kind; float ax, ay, az, bx, by, bz, scalar; uint32_t w, h; uint64_t offset; };

// Work item: one 8x8 tile, the lanes (pixels) of it that are to be rendered, and where its pixels go.
struct WorkItem { uint32_t tile; uint32_t out_tile; uint64_t mask; };

struct KernelArgs {
    const float4 *nodes;          // 2 x float4 per crt_node
    const uint32_t *leaf_tris;
    const uint32_t *leaf_meshes;
    const float4 *tris;           // 4 x float4 per crt_triangle
    const uint32_t *tri_verts;    // 3 per triangle
    const float *vnormals;        // 3 per vertex
    const float *vuvs;            // 3 per vertex (or null)
    const crt_mesh *meshes;
    const DMaterial *materials;
    const DTexture *textures;
    const uint32_t *texels;       // RGBX8
    const float4 *lights;         // xyz + (float)intensity
    uint32_t n_lights, top_root;
    float bgx, bgy, bgz;
    uint32_t width, height, tiles_x;
    float cam_pos[3];
    float cam[9];
    uint32_t max_depth;
    float shadow_bias, reflection_bias, refraction_bias;
    const WorkItem *items;
    uint32_t n_items;
    uint32_t *pixel_counter;      // next unassigned (item*64 + lane)
    float *out;                   // frame (row major) or packed tiles
    uint32_t packed;              // 0: out is the H*W*3 frame, 1: out is packed by out_tile
    float *frames;                // [wave][level][FRAME_DWORDS][64]
    uint64_t frame_wave_stride;   // floats per wave
    unsigned long long *counters; // C_N, counting build only
};

// ---------------------------------------------------------------------------------------------
// exact-arithmetic helpers (expression shapes follow Vector.cpp; compiled with -ffp-contract=off)
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;  // Vector.cpp:57-59
}
__device__ __forceinline__ float len3(float x, float y, float z) {
    return sqrtf(x * x + y * y + z * z);  // Vector.cpp:114-117
}
__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {  // Vector.cpp:97-106
    float length = len3(x, y, z);
    if (length == 0) return;
    length = 1.0f / length;
    x *= length; y *= length; z *= length;
}
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }

struct Lane {
    // current ray
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    uint32_t parmask;   // bit i: |d_i| < FLT_EPSILON (BoundingBox.h:90)
    int rtype;
    // traversal cursors
    uint32_t tnode, tleaf, mnode, mleaf, cur_mesh;
    // mesh-level and scene-level running closest hit (KDTree.cpp:75-86, 156-167)
    bool mhave, have, occluded;
    float mmin, mt, tmin, bt, light_dist;
    uint32_t mtri, btri, bmesh;
};

__device__ __forceinline__ void ray_prepare(Lane &L) {
    // 1/d is recomputed per box in the reference (BoundingBox.h:95); it only depends on the ray.
    L.parmask = (fabsf(L.dx) < FLT_EPSILON ? 1u : 0u) | (fabsf(L.dy) < FLT_EPSILON ? 2u : 0u) |
                (fabsf(L.dz) < FLT_EPSILON ? 4u : 0u);
    L.ix = 1.0f / L.dx;
    L.iy = 1.0f / L.dy;
    L.iz = 1.0f / L.dz;
}

// BoundingBox::hasIntersection (BoundingBox.h:85-108).  The reference returns early per axis; t0 only
// grows and t1 only shrinks (NaNs are never selected by std::max/std::min as written), so testing
// t0 > t1 once at the end gives the same verdict.
__device__ __forceinline__ bool slab_test(const Lane &L, const float4 &q0, const float4 &q1) {
    float t0 = -FLT_MAX, t1 = FLT_MAX;
    bool reject = false;
    {
        float tn = (q0.x - L.ox) * L.ix, tf = (q1.x - L.ox) * L.ix;
        if (tn > tf) { float s = tn; tn = tf; tf = s; }
        if (L.parmask & 1u) reject |= (L.ox < q0.x) || (L.ox > q1.x);
        else { t0 = std_max(t0, tn); t1 = std_min(t1, tf); }
    }
    {
        float tn = (q0.y - L.oy) * L.iy, tf = (q1.y - L.oy) * L.iy;
        if (tn > tf) { float s = tn; tn = tf; tf = s; }
        if (L.parmask & 2u) reject |= (L.oy < q0.y) || (L.oy > q1.y);
        else { t0 = std_max(t0, tn); t1 = std_min(t1, tf); }
    }
    {
        float tn = (q0.z - L.oz) * L.iz, tf = (q1.z - L.oz) * L.iz;
        if (tn > tf) { float s = tn; tn = tf; tf = s; }
        if (L.parmask & 4u) reject |= (L.oz < q0.z) || (L.oz > q1.z);
        else { t0 = std_max(t0, tn); t1 = std_min(t1, tf); }
    }
    return !(reject || (t0 > t1));
}

// Ray::intersectWithTriangle + Triangle::pointIsInTriangle (Ray.cpp:9-31, Triangle.cpp:37-57).
__device__ __forceinline__ bool triangle_test(const Lane &L, const float4 &a, const float4 &b, const float4 &c,
                                              float plane, float &t_out) {
    const float nx = a.w, ny = b.w, nz = c.w;
    const float nd = dot3(L.dx, L.dy, L.dz, nx, ny, nz);
    if (L.rtype == RAY_PRIMARY && nd >= 0) return false;
    const float t = -(dot3(nx, ny, nz, L.ox, L.oy, L.oz) + plane) / nd;
    if (t < 0) return false;
    const float px = L.ox + L.dx * t, py = L.oy + L.dy * t, pz = L.oz + L.dz * t;
    {
        const float ex = b.x - a.x, ey = b.y - a.y, ez = b.z - a.z;
        const float cx = px - a.x, cy = py - a.y, cz = pz - a.z;
        if (dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx) < -FLT_EPSILON) return false;
    }
    {
        const float ex = c.x - b.x, ey = c.y - b.y, ez = c.z - b.z;
        const float cx = px - b.x, cy = py - b.y, cz = pz - b.z;
        if (dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx) < -FLT_EPSILON) return false;
    }
    {
        const float ex = a.x - c.x, ey = a.y - c.y, ez = a.z - c.z;
        const float cx = px - c.x, cy = py - c.y, cz = pz - c.z;
        if (dot3(nx, ny, nz, ey * cz - ez * cy, ez * cx - ex * cz, ex * cy - ey * cx) < -FLT_EPSILON) return false;
    }
    t_out = t;
    return true;
}

__device__ __forceinline__ void traversal_begin(Lane &L, uint32_t top_root) {
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
}

// One unit of traversal work for this lane: either one triangle test (when inside a leaf) or one
// node visit / bookkeeping step.  Returns false when the whole two-level walk is finished.
template <bool COUNT>
__device__ __forceinline__ bool traversal_step(Lane &L, const KernelArgs &A, uint32_t *cnt) {
    if (L.mleaf != NONE) {
        // ---- inside a mesh-tree leaf: test one triangle (KDTree.cpp:57-65)
        const uint32_t ent = A.leaf_tris[L.mleaf];
        const uint32_t tri = ent & ~LAST;
        L.mleaf = (ent & LAST) ? NONE : L.mleaf + 1;
        const float4 a = A.tris[4 * (size_t)tri + 0];
        const float4 b = A.tris[4 * (size_t)tri + 1];
        const float4 c = A.tris[4 * (size_t)tri + 2];
        const float plane = A.tris[4 * (size_t)tri + 3].x;
        if (COUNT) { cnt[C_TRI]++; cnt[C_LEAFIDX]++; }
        float t;
        if (triangle_test(L, a, b, c, plane, t)) {
            // `closest = hits[0]; min = inf; for h: if (h.d < min) {min = h.d; closest = h}` fused into the walk
            if (!L.mhave) { L.mhave = true; L.mt = t; L.mtri = tri; }
            if (t < L.mmin) { L.mmin = t; L.mt = t; L.mtri = tri; }
        }
        return true;
    }
    if (L.cur_mesh != NONE) {
        if (L.mnode != END) {
            // ---- visit one mesh-tree node (KDTree.cpp:53-74)
            const float4 q0 = A.nodes[2 * (size_t)L.mnode], q1 = A.nodes[2 * (size_t)L.mnode + 1];
            const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
            if (COUNT) cnt[C_BOX]++;
            const bool hit = slab_test(L, q0, q1);
            if (hit && (link & LEAF)) {
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
                const float px = L.ox + L.dx * L.mt, py = L.oy + L.dy * L.mt, pz = L.oz + L.dz * L.mt;
                if (len3(px - L.ox, py - L.oy, pz - L.oz) <= L.light_dist) L.occluded = true;
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
        const uint32_t ent = A.leaf_meshes[L.tleaf];
        const uint32_t mi = ent & ~LAST;
        L.tleaf = (ent & LAST) ? NONE : L.tleaf + 1;
        if (COUNT) cnt[C_LEAFIDX]++;
        const crt_mesh m = A.meshes[mi];
        if (L.rtype == RAY_SHADOW && (m.flags & 1u)) return true;
        L.cur_mesh = mi;
        L.mnode = m.root;
        L.mhave = false;
        L.mmin = INFINITY;
        return true;
    }
    if (L.tnode != END) {
        // ---- visit one top-level node (KDTree.cpp:132-155)
        const float4 q0 = A.nodes[2 * (size_t)L.tnode], q1 = A.nodes[2 * (size_t)L.tnode + 1];
        const uint32_t miss = __float_as_uint(q0.w), link = __float_as_uint(q1.w);
        if (COUNT) cnt[C_BOX]++;
        const bool hit = slab_test(L, q0, q1);
        if (hit && (link & LEAF)) {
            L.tleaf = link & ~LEAF;
            L.tnode = miss;
        } else {
            L.tnode = hit ? link : miss;
        }
        return true;
    }
    return false;
}

// Texture::getColor (Texture.cpp:14-72)
template <bool COUNT>
__device__ __forceinline__ void texture_color(const KernelArgs &A, const DTexture &T, uint32_t tri, float u, float v,
                                              float w, float &r, float &g, float &b, bool &is_bitmap) {
    is_bitmap = false;
    if (T.kind == CRT_TEX_ALBEDO) { r = T.ax; g = T.ay; b = T.az; return; }
    if (T.kind == CRT_TEX_EDGES) {
        if (u < T.scalar || v < T.scalar || w < T.scalar) { r = T.bx; g = T.by; b = T.bz; }
        else { r = T.ax; g = T.ay; b = T.az; }
        return;
    }
    const uint32_t i0 = A.tri_verts[3 * (size_t)tri], i1 = A.tri_verts[3 * (size_t)tri + 1],
                   i2 = A.tri_verts[3 * (size_t)tri + 2];
    // u * UV1 + v * UV2 + (w * UV0), Texture.cpp:34-36 / 63-65 (only x and y are used)
    const float uvx = (u * A.vuvs[3 * (size_t)i1] + v * A.vuvs[3 * (size_t)i2]) + w * A.vuvs[3 * (size_t)i0];
    const float uvy = (u * A.vuvs[3 * (size_t)i1 + 1] + v * A.vuvs[3 * (size_t)i2 + 1]) + w * A.vuvs[3 * (size_t)i0 + 1];
    if (T.kind == CRT_TEX_CHECKER) {
        const unsigned int x = (unsigned int)(uvx / T.scalar);
        const unsigned int y = (unsigned int)(uvy / T.scalar);
        if (x % 2 == y % 2) { r = T.ax; g = T.ay; b = T.az; } else { r = T.bx; g = T.by; b = T.bz; }
        return;
    }
    is_bitmap = true;
    int x = (int)(uvx * (float)(int)T.w);
    int y = (int)((1.0f - uvy) * (float)(int)T.h);
    x = (x < 0) ? 0 : (((int)T.w - 1 < x) ? (int)T.w - 1 : x);  // std::clamp
    y = (y < 0) ? 0 : (((int)T.h - 1 < y) ? (int)T.h - 1 : y);
    const uint32_t px = A.texels[T.offset + (size_t)y * T.w + (size_t)x];
    const float coefficient = 1.0f / 255.0f;  // Texture.cpp:53-57
    r = (float)(px & 255u) * coefficient;
    g = (float)((px >> 8) & 255u) * coefficient;
    b = (float)((px >> 16) & 255u) * coefficient;
}

template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void render_kernel(const KernelArgs A) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    float *frames = A.frames + (size_t)wave * A.frame_wave_stride + lane;
    auto FR = [&](uint32_t level, int field) -> float & { return frames[((size_t)level * FRAME_DWORDS + field) * 64]; };
    auto FRK = [&](uint32_t level) -> int & { return *reinterpret_cast<int *>(&frames[(size_t)level * FRAME_DWORDS * 64]); };

    uint32_t cnt[C_N];
    if (COUNT) for (int k = 0; k < C_N; k++) cnt[k] = 0;

    Lane L;
    int state = ST_FETCH;
    uint32_t sp = 0;           // recursion level of the current ray == number of frames below it
    size_t out_off = 0;
    // diffuse light loop (RayTracer.cpp:300-330)
    float hpx = 0, hpy = 0, hpz = 0, hnx = 0, hny = 0, hnz = 0;
    float basex = 0, basey = 0, basez = 0, accx = 0, accy = 0, accz = 0, kfac = 0;
    uint32_t li = 0;
    bool base_is_bitmap = false;
    const uint32_t total_px = A.n_items * 64u;

    for (;;) {
        // ------------------------------------------------------------------ fetch new pixels
        if (__ballot(state == ST_FETCH)) {
            while (state == ST_FETCH) {
                const unsigned long long need = __ballot(1);
                const int n = __popcll(need);
                const int rank = __popcll(need & ((1ull << lane) - 1ull));
                uint32_t base = 0;
                if (rank == 0) base = atomicAdd(A.pixel_counter, (uint32_t)n);
                base = __shfl(base, __ffsll((long long)need) - 1);
                const uint32_t q = base + (uint32_t)rank;
                if (q >= total_px) { state = ST_DONE; break; }
                const WorkItem wi = A.items[q >> 6];
                const uint32_t sub = q & 63u;
                const uint32_t px = (wi.tile % A.tiles_x) * TILE + (sub & 7u);
                const uint32_t py = (wi.tile / A.tiles_x) * TILE + (sub >> 3);
                if (!((wi.mask >> sub) & 1ull) || px >= A.width || py >= A.height) continue;  // not covered: take another
                out_off = A.packed ? ((size_t)wi.out_tile * 64 + sub) * 3 : ((size_t)py * A.width + px) * 3;
                // RayTracer::getRay (RayTracer.cpp:61-80), pixel centre
                float x = (float)px + 0.5f;
                float y = (float)py + 0.5f;
                x = x / (float)A.width;
                y = y / (float)A.height;
                x = (2.0f * x) - 1.0f;
                y = 1.0f - (2.0f * y);
                x = x * ((float)A.width / (float)A.height);
                const float z = -1.0f;
                L.dx = x * A.cam[0] + y * A.cam[3] + z * A.cam[6];   // row vector x matrix, Matrix.h:137-142
                L.dy = x * A.cam[1] + y * A.cam[4] + z * A.cam[7];
                L.dz = x * A.cam[2] + y * A.cam[5] + z * A.cam[8];
                normalize3(L.dx, L.dy, L.dz);
                L.ox = A.cam_pos[0]; L.oy = A.cam_pos[1]; L.oz = A.cam_pos[2];
                L.rtype = RAY_PRIMARY;
                sp = 0;
                // shootRay entry (RayTracer.cpp:420-429): normalise again, depth 0 <= MAX_DEPTH always
                normalize3(L.dx, L.dy, L.dz);
                if (COUNT) cnt[C_PRIMARY]++;
                ray_prepare(L);
                traversal_begin(L, A.top_root);
                state = ST_TRAVERSE;
            }
        }
        if (!__ballot(state != ST_DONE)) break;

        // ------------------------------------------------------------------ traverse
        bool finished = false;
        if (state == ST_TRAVERSE) {
            int budget = 48;
            do {
                if (!traversal_step<COUNT>(L, A, cnt)) { finished = true; break; }
            } while (--budget > 0);
        }

        // ------------------------------------------------------------------ a walk ended: shade / continue
        if (finished) {
            bool returning = false;      // a colour is being returned to the caller level
            bool new_ray = false;        // L holds a new ray that enters shootRay at level sp
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
                cx = A.bgx; cy = A.bgy; cz = A.bgz; returning = true;  // RayTracer.cpp:449-450
            } else {
                // ---- closest hit: KDTree.cpp:168-190
                const float4 ta = A.tris[4 * (size_t)L.btri + 0], tb = A.tris[4 * (size_t)L.btri + 1],
                             tc = A.tris[4 * (size_t)L.btri + 2];
                const float px = L.ox + L.dx * L.bt, py = L.oy + L.dy * L.bt, pz = L.oz + L.dz * L.bt;
                float nx = ta.w, ny = tb.w, nz = tc.w;
                const DMaterial M = A.materials[A.meshes[L.bmesh].material];
                float u = 0, v = 0;
                if (COUNT) cnt[C_HIT]++;
                if (M.smooth || M.texture >= 0) {
                    // Triangle::getBarycentricCoordinates (Triangle.cpp:63-73)
                    const float v0px = px - ta.x, v0py = py - ta.y, v0pz = pz - ta.z;
                    const float e1x = tb.x - ta.x, e1y = tb.y - ta.y, e1z = tb.z - ta.z;
                    const float e2x = tc.x - ta.x, e2y = tc.y - ta.y, e2z = tc.z - ta.z;
                    const float area = len3(e1y * e2z - e1z * e2y, e1z * e2x - e1x * e2z, e1x * e2y - e1y * e2x);
                    u = len3(v0py * e2z - v0pz * e2y, v0pz * e2x - v0px * e2z, v0px * e2y - v0py * e2x) / area;
                    v = len3(e1y * v0pz - e1z * v0py, e1z * v0px - e1x * v0pz, e1x * v0py - e1y * v0px) / area;
                    if (M.smooth) {
                        const uint32_t i0 = A.tri_verts[3 * (size_t)L.btri], i1 = A.tri_verts[3 * (size_t)L.btri + 1],
                                       i2 = A.tri_verts[3 * (size_t)L.btri + 2];
                        const float w = 1 - u - v;
                        nx = (A.vnormals[3 * (size_t)i1] * u + A.vnormals[3 * (size_t)i2] * v) + A.vnormals[3 * (size_t)i0] * w;
                        ny = (A.vnormals[3 * (size_t)i1 + 1] * u + A.vnormals[3 * (size_t)i2 + 1] * v) + A.vnormals[3 * (size_t)i0 + 1] * w;
                        nz = (A.vnormals[3 * (size_t)i1 + 2] * u + A.vnormals[3 * (size_t)i2 + 2] * v) + A.vnormals[3 * (size_t)i0 + 2] * w;
                        normalize3(nx, ny, nz);
                    }
                }
                if (M.type == CRT_MAT_DIFFUSE) {
                    hpx = px; hpy = py; hpz = pz; hnx = nx; hny = ny; hnz = nz;
                    base_is_bitmap = false;
                    if (M.texture >= 0) {
                        texture_color<COUNT>(A, A.textures[M.texture], L.btri, u, v, 1.0f - u - v, basex, basey, basez,
                                             base_is_bitmap);
                    } else { basex = M.ax; basey = M.ay; basez = M.az; }
                    accx = accy = accz = 0;
                    li = 0;
                    next_light = true;
                } else if (M.type == CRT_MAT_REFLECTIVE) {
                    // RayTracer::calculateReflection (RayTracer.cpp:358-374)
                    FRK(sp) = FR_REFLECT;
                    FR(sp, 1) = M.ax; FR(sp, 2) = M.ay; FR(sp, 3) = M.az;
                    const float k = 2 * dot3(L.dx, L.dy, L.dz, nx, ny, nz);  // Vector::reflect, Vector.cpp:119-122
                    const float rx = L.dx - k * nx, ry = L.dy - k * ny, rz = L.dz - k * nz;
                    L.ox = px + nx * A.reflection_bias; L.oy = py + ny * A.reflection_bias; L.oz = pz + nz * A.reflection_bias;
                    L.dx = rx; L.dy = ry; L.dz = rz;
                    normalize3(L.dx, L.dy, L.dz);
                    L.rtype = RAY_REFLECTION;
                    sp++;
                    new_ray = true;
                } else if (M.type == CRT_MAT_REFRACTIVE) {
                    // RayTracer::calculateRefraction (RayTracer.cpp:375-417)
                    float eta1 = 1.0f, eta2 = M.ior;
                    float idn = dot3(L.dx, L.dy, L.dz, nx, ny, nz);
                    if (idn > 0) {
                        const float s = eta1; eta1 = eta2; eta2 = s;
                        nx = -1.0f * nx; ny = -1.0f * ny; nz = -1.0f * nz;
                        idn = -idn;
                    }
                    const float cos_a = -idn;
                    const float sin_a = sqrtf(std_max(0.0f, 1 - cos_a * cos_a));
                    const float k = 2 * dot3(L.dx, L.dy, L.dz, nx, ny, nz);
                    const float rx = L.dx - k * nx, ry = L.dy - k * ny, rz = L.dz - k * nz;
                    const float eta_ratio = eta1 / eta2;
                    const float sin_b = eta_ratio * sin_a;
                    if (sin_b < 1.0f) {
                        const float q = (eta1 - eta2) / (eta1 + eta2);
                        const float r0 = q * q;  // std::powf(q, 2): folded to q*q by the reference's compiler at -O2
                        const float fresnel = r0 + (1 - r0) * crt_pow5(1.0f - cos_a);
                        const float cos_b = sqrtf(std_max(0.0f, 1 - sin_b * sin_b));
                        float tx = eta_ratio * (L.dx + cos_a * nx) - cos_b * nx;
                        float ty = eta_ratio * (L.dy + cos_a * ny) - cos_b * ny;
                        float tz = eta_ratio * (L.dz + cos_a * nz) - cos_b * nz;
                        normalize3(tx, ty, tz);
                        FRK(sp) = FR_REFRACT_WAIT_REFLECTION;
                        FR(sp, 1) = px - nx * A.refraction_bias; FR(sp, 2) = py - ny * A.refraction_bias;
                        FR(sp, 3) = pz - nz * A.refraction_bias;
                        FR(sp, 4) = tx; FR(sp, 5) = ty; FR(sp, 6) = tz;
                        FR(sp, 7) = fresnel;
                    } else {
                        FRK(sp) = FR_REFRACT_NO_TRANSMISSION;
                    }
                    L.ox = px + nx * A.reflection_bias; L.oy = py + ny * A.reflection_bias; L.oz = pz + nz * A.reflection_bias;
                    L.dx = rx; L.dy = ry; L.dz = rz;
                    normalize3(L.dx, L.dy, L.dz);
                    L.rtype = RAY_REFLECTION;
                    sp++;
                    new_ray = true;
                } else {
                    cx = A.bgx; cy = A.bgy; cz = A.bgz; returning = true;  // RayTracer.cpp:443-446
                }
            }

            // ---- diffuse light loop: set up the next shadow ray or return the accumulated colour
            if (next_light) {
                if (li < A.n_lights) {
                    const float4 lg = A.lights[li];
                    if (COUNT) { cnt[C_LIGHT]++; cnt[C_SHADOW]++; }
                    float lx = lg.x - hpx, ly = lg.y - hpy, lz = lg.z - hpz;
                    const float dist = len3(lx, ly, lz);
                    const float area = 4 * dist * dist * PI_F;
                    normalize3(lx, ly, lz);
                    const float angle = std_max(0.0f, dot3(lx, ly, lz, hnx, hny, hnz));
                    kfac = lg.w / area * angle;
                    L.ox = hpx + hnx * A.shadow_bias; L.oy = hpy + hny * A.shadow_bias; L.oz = hpz + hnz * A.shadow_bias;
                    L.dx = lx; L.dy = ly; L.dz = lz;
                    L.rtype = RAY_SHADOW;
                    L.light_dist = dist;
                    ray_prepare(L);
                    traversal_begin(L, A.top_root);
                } else {
                    cx = accx; cy = accy; cz = accz; returning = true;
                }
            }

            // ---- unwind / advance the explicit recursion (post-order, as the reference's call stack does)
            while (returning || new_ray) {
                if (new_ray) {
                    // shootRay entry (RayTracer.cpp:419-429)
                    normalize3(L.dx, L.dy, L.dz);
                    new_ray = false;
                    if (sp > A.max_depth) { cx = A.bgx; cy = A.bgy; cz = A.bgz; returning = true; continue; }
                    if (COUNT) cnt[C_SECONDARY]++;
                    ray_prepare(L);
                    traversal_begin(L, A.top_root);
                    break;
                }
                if (sp == 0) {
                    A.out[out_off] = cx; A.out[out_off + 1] = cy; A.out[out_off + 2] = cz;  // RayTracer.cpp:106
                    state = ST_FETCH;
                    break;
                }
                const uint32_t f = sp - 1;
                const int kind = FRK(f);
                if (kind == FR_REFLECT) {
                    cx = 0.0f + FR(f, 1) * cx; cy = 0.0f + FR(f, 2) * cy; cz = 0.0f + FR(f, 3) * cz;  // RayTracer.cpp:368-372
                    sp = f;
                } else if (kind == FR_REFRACT_NO_TRANSMISSION) {
                    sp = f;  // `return reflectionColor`, RayTracer.cpp:416
                } else if (kind == FR_REFRACT_WAIT_REFLECTION) {
                    L.ox = FR(f, 1); L.oy = FR(f, 2); L.oz = FR(f, 3);
                    L.dx = FR(f, 4); L.dy = FR(f, 5); L.dz = FR(f, 6);
                    L.rtype = RAY_REFRACTION;
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

// scatter gathered packed tiles into the row-major frame
__global__ void unpack_kernel(const float *packed, uint32_t n_parts, uint64_t part_stride, float *frame, uint32_t width,
                              uint32_t height, uint32_t tiles_x, uint32_t n_tiles) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (tile, pixel)
    if (gid >= (uint64_t)n_tiles * 64) return;
    const uint32_t tile = (uint32_t)(gid >> 6), sub = (uint32_t)(gid & 63);
    const uint32_t px = (tile % tiles_x) * TILE + (sub & 7u), py = (tile / tiles_x) * TILE + (sub >> 3);
    if (px >= width || py >= height) return;
    const uint32_t part = tile % n_parts, local = tile / n_parts;
    const float *src = packed + (uint64_t)part * part_stride + ((uint64_t)local * 64 + sub) * 3;
    float *dst = frame + ((uint64_t)py * width + px) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// PPMColor (Color.cpp:12-16): (unsigned short)(std::clamp(c, 0.0f, 1.0f) * 255)
__global__ void quantize_kernel(const float *rgb, uint64_t n, uint8_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = rgb[i];
    const float cl = (c < 0.0f) ? 0.0f : ((1.0f < c) ? 1.0f : c);
    out[i] = (uint8_t)(unsigned short)(cl * 255);
}

}  // namespace

// =================================================================================================
// host side of the C ABI
struct crt_ctx {
    int device = 0;
    std::string error;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0;
    KernelArgs args{};
    std::vector<void *> allocs;
    float *d_frame = nullptr;
    uint8_t *d_quant = nullptr;
    WorkItem *d_items = nullptr;
    size_t items_cap = 0;
    std::vector<crt_rect> cached_rects;
    uint32_t cached_n_items = 0;
    uint64_t cached_pixels = 0;
    uint32_t *d_pixel_counter = nullptr;
    unsigned long long *d_counters = nullptr;
    float *d_frames = nullptr;
    size_t frames_floats = 0;
    uint32_t grid_blocks = 0;
    crt_stats stats{};
    int num_cus = 0;
};

static std::string g_create_error;

#define CRT_HIP_CHECK(ctx, expr)                                                                    \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return CRT_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

template <typename T>
static int upload(crt_ctx *ctx, const T *src, size_t count, const T **dst) {
    void *p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    CRT_HIP_CHECK(ctx, hipMalloc(&p, bytes));
    ctx->allocs.push_back(p);
    if (count) CRT_HIP_CHECK(ctx, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)p;
    return CRT_OK;
}

extern "C" int crt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int validate_scene(const crt_scene_desc *s, std::string &err) {
    auto bad = [&](const char *m) { err = m; return CRT_ERR_INVALID; };
    if (!s) return bad("scene is NULL");
    if (s->width == 0 || s->height == 0) return bad("empty image");
    if (!s->nodes || s->n_nodes == 0 || s->top_root >= s->n_nodes) return bad("missing tree nodes");
    if (s->n_triangles && (!s->triangles || !s->triangle_vertices)) return bad("missing triangle arrays");
    if (s->n_vertices && !s->vertex_normals) return bad("missing vertex normals");
    if (s->n_meshes && !s->meshes) return bad("missing meshes");
    if (s->n_materials == 0 && s->n_meshes) return bad("meshes without materials");
    if (s->n_leaf_triangles > 0x7FFFFFFFull) return bad("too many leaf entries");
    // every index the kernel will follow is checked here, on the host, before anything is launched
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const crt_node &n = s->nodes[i];
        if (n.miss != CRT_LINK_END && n.miss >= s->n_nodes) return bad("node miss link out of range");
        if (n.link & CRT_LINK_LEAF) continue;
        if (n.link != CRT_LINK_END && n.link >= s->n_nodes) return bad("node hit link out of range");
    }
    for (uint64_t i = 0; i < s->n_leaf_triangles; i++)
        if ((s->leaf_triangles[i] & ~CRT_ENTRY_LAST) >= s->n_triangles) return bad("leaf triangle index out of range");
    if (s->n_leaf_triangles && !(s->leaf_triangles[s->n_leaf_triangles - 1] & CRT_ENTRY_LAST)) return bad("unterminated triangle leaf");
    for (uint32_t i = 0; i < s->n_leaf_meshes; i++)
        if ((s->leaf_meshes[i] & ~CRT_ENTRY_LAST) >= s->n_meshes) return bad("leaf mesh index out of range");
    if (s->n_leaf_meshes && !(s->leaf_meshes[s->n_leaf_meshes - 1] & CRT_ENTRY_LAST)) return bad("unterminated mesh leaf");
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        if (s->meshes[i].root >= s->n_nodes) return bad("mesh root out of range");
        if (s->meshes[i].material >= s->n_materials) return bad("mesh material out of range");
    }
    for (uint64_t i = 0; i < (uint64_t)s->n_triangles * 3; i++)
        if (s->triangle_vertices[i] >= s->n_vertices) return bad("triangle vertex index out of range");
    bool any_uv_texture = false;
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const crt_material &m = s->materials[i];
        if (m.texture >= 0) {
            if ((uint32_t)m.texture >= s->n_textures) return bad("material texture out of range");
            uint32_t k = s->textures[m.texture].kind;
            if (k == CRT_TEX_CHECKER || k == CRT_TEX_BITMAP) any_uv_texture = true;
        }
    }
    if (any_uv_texture && !s->vertex_uvs) return bad("textured material without vertex uvs");
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const crt_texture &t = s->textures[i];
        if (t.kind > CRT_TEX_BITMAP) return bad("unknown texture kind");
        if (t.kind == CRT_TEX_BITMAP) {
            if (t.width == 0 || t.height == 0) return bad("empty bitmap");
            if (t.texel_offset + (uint64_t)t.width * t.height > s->n_texels) return bad("bitmap texels out of range");
        }
    }
    // leaf links must point inside the entry arrays; a leaf of the top tree is told from a mesh-tree
    // leaf by reachability, so both bounds are checked against the larger array here and the
    // builder's own tests pin the exact structure
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const crt_node &n = s->nodes[i];
        if (!(n.link & CRT_LINK_LEAF)) continue;
        uint64_t b = n.link & ~CRT_LINK_LEAF;
        if (b >= s->n_leaf_triangles && b >= s->n_leaf_meshes) return bad("leaf begin out of range");
    }
    return CRT_OK;
}

extern "C" int crt_create(const crt_scene_desc *s, int device, crt_ctx **out) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    int rc = validate_scene(s, g_create_error);
    if (rc != CRT_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (this library has no CPU fallback)";
        return CRT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return CRT_ERR_NO_DEVICE;
    }
    crt_ctx *ctx = new (std::nothrow) crt_ctx();
    if (!ctx) return CRT_ERR_NOMEM;
    ctx->device = device;
    auto fail = [&](int code) {
        g_create_error = ctx->error;
        crt_destroy(ctx);
        return code;
    };
#define CK(expr)                                                                  \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            return fail(CRT_ERR_HIP);                                             \
        }                                                                         \
    } while (0)
    CK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount;
    CK(hipStreamCreate(&ctx->stream));
    CK(hipEventCreate(&ctx->ev0));
    CK(hipEventCreate(&ctx->ev1));

    ctx->width = s->width;
    ctx->height = s->height;
    ctx->tiles_x = (s->width + TILE - 1) / TILE;
    ctx->tiles_y = (s->height + TILE - 1) / TILE;
    KernelArgs &A = ctx->args;
    static_assert(sizeof(crt_node) == 32 && sizeof(crt_triangle) == 64, "record sizes");
    if (upload(ctx, (const float4 *)s->nodes, (size_t)s->n_nodes * 2, &A.nodes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_triangles, (size_t)s->n_leaf_triangles, &A.leaf_tris)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_meshes, (size_t)s->n_leaf_meshes, &A.leaf_meshes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, (const float4 *)s->triangles, (size_t)s->n_triangles * 4, &A.tris)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->triangle_vertices, (size_t)s->n_triangles * 3, &A.tri_verts)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->vertex_normals, (size_t)s->n_vertices * 3, &A.vnormals)) return fail(CRT_ERR_HIP);
    if (s->vertex_uvs) {
        if (upload(ctx, s->vertex_uvs, (size_t)s->n_vertices * 3, &A.vuvs)) return fail(CRT_ERR_HIP);
    } else A.vuvs = nullptr;
    if (upload(ctx, s->meshes, (size_t)s->n_meshes, &A.meshes)) return fail(CRT_ERR_HIP);
    {
        std::vector<DMaterial> mats(s->n_materials);
        for (uint32_t i = 0; i < s->n_materials; i++) {
            const crt_material &m = s->materials[i];
            mats[i] = DMaterial{m.albedo[0], m.albedo[1], m.albedo[2], m.ior, m.type, m.smooth, m.texture, 0};
        }
        if (upload(ctx, mats.data(), mats.size(), &A.materials)) return fail(CRT_ERR_HIP);
        std::vector<DTexture> tex(s->n_textures);
        for (uint32_t i = 0; i < s->n_textures; i++) {
            const crt_texture &t = s->textures[i];
            tex[i] = DTexture{t.kind, t.color_a[0], t.color_a[1], t.color_a[2], t.color_b[0], t.color_b[1], t.color_b[2],
                              t.scalar, t.width, t.height, t.texel_offset};
        }
        if (upload(ctx, tex.data(), tex.size(), &A.textures)) return fail(CRT_ERR_HIP);
        std::vector<uint32_t> px((size_t)s->n_texels);
        for (uint64_t i = 0; i < s->n_texels; i++)
            px[i] = (uint32_t)s->texels[3 * i] | ((uint32_t)s->texels[3 * i + 1] << 8) | ((uint32_t)s->texels[3 * i + 2] << 16);
        if (upload(ctx, px.data(), px.size(), &A.texels)) return fail(CRT_ERR_HIP);
        std::vector<float4> lights(s->n_lights);
        for (uint32_t i = 0; i < s->n_lights; i++)
            lights[i] = make_float4(s->lights[i].position[0], s->lights[i].position[1], s->lights[i].position[2],
                                    (float)s->lights[i].intensity);  // static_cast<float>(light.intentsity), RayTracer.cpp:320
        if (upload(ctx, lights.data(), lights.size(), &A.lights)) return fail(CRT_ERR_HIP);
    }
    A.n_lights = s->n_lights;
    A.top_root = s->top_root;
    A.bgx = s->background[0]; A.bgy = s->background[1]; A.bgz = s->background[2];
    A.width = s->width; A.height = s->height; A.tiles_x = ctx->tiles_x;
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(A.cam, ident, sizeof(ident));
    A.cam_pos[0] = A.cam_pos[1] = A.cam_pos[2] = 0;

    size_t frame_bytes = (size_t)s->width * s->height * 3 * sizeof(float);
    CK(hipMalloc((void **)&ctx->d_frame, frame_bytes));
    CK(hipMemset(ctx->d_frame, 0, frame_bytes));  // colorBuffer starts as Color() = (0,0,0), RayTracer.cpp:46-50
    CK(hipMalloc((void **)&ctx->d_quant, (size_t)s->width * s->height * 3));
    CK(hipMalloc((void **)&ctx->d_pixel_counter, sizeof(uint32_t)));
    CK(hipMalloc((void **)&ctx->d_counters, C_N * sizeof(unsigned long long)));
    // persistent grid: 8 blocks of 256 threads per CU gives every CU its 32 waves if registers allow
    ctx->grid_blocks = (uint32_t)ctx->num_cus * 8u;
#undef CK
    *out = ctx;
    return CRT_OK;
}

extern "C" void crt_destroy(crt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (void *p : ctx->allocs) (void)hipFree(p);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    if (ctx->d_quant) (void)hipFree(ctx->d_quant);
    if (ctx->d_items) (void)hipFree(ctx->d_items);
    if (ctx->d_pixel_counter) (void)hipFree(ctx->d_pixel_counter);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *crt_last_error(const crt_ctx *ctx) {
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

extern "C" int crt_set_camera(crt_ctx *ctx, const float position[3], const float matrix[9]) {
    if (!ctx || !position || !matrix) return CRT_ERR_INVALID;
    memcpy(ctx->args.cam_pos, position, 3 * sizeof(float));
    memcpy(ctx->args.cam, matrix, 9 * sizeof(float));
    return CRT_OK;
}

static int check_options(crt_ctx *ctx, const crt_options *o) {
    if (!o) { ctx->error = "options is NULL"; return CRT_ERR_INVALID; }
    if (o->use_gi) { ctx->error = "USE_GI is outside this path (non-deterministic in the reference)"; return CRT_ERR_INVALID; }
    if (o->max_depth > 4096) { ctx->error = "max_depth too large"; return CRT_ERR_INVALID; }
    return CRT_OK;
}

static int ensure_items(crt_ctx *ctx, size_t n) {
    if (n <= ctx->items_cap) return CRT_OK;
    if (ctx->d_items) (void)hipFree(ctx->d_items);
    ctx->d_items = nullptr;
    ctx->items_cap = 0;
    CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_items, n * sizeof(WorkItem)));
    ctx->items_cap = n;
    ctx->cached_rects.clear();
    return CRT_OK;
}

static int ensure_frames(crt_ctx *ctx, uint32_t max_depth) {
    const size_t waves = (size_t)ctx->grid_blocks * (BLOCK / 64);
    const size_t per_wave = (size_t)(max_depth + 1) * FRAME_DWORDS * 64;
    if (waves * per_wave > ctx->frames_floats) {
        if (ctx->d_frames) (void)hipFree(ctx->d_frames);
        ctx->d_frames = nullptr;
        ctx->frames_floats = 0;
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_frames, waves * per_wave * sizeof(float)));
        ctx->frames_floats = waves * per_wave;
    }
    ctx->args.frames = ctx->d_frames;
    ctx->args.frame_wave_stride = per_wave;
    return CRT_OK;
}

static int launch_render(crt_ctx *ctx, const crt_options *o, uint32_t n_items, float *d_out, uint32_t packed,
                         hipStream_t stream, bool timed) {
    int rc = ensure_frames(ctx, o->max_depth);
    if (rc) return rc;
    KernelArgs &A = ctx->args;
    A.max_depth = o->max_depth;
    A.shadow_bias = o->shadow_bias;
    A.reflection_bias = o->reflection_bias;
    A.refraction_bias = o->refraction_bias;
    A.items = ctx->d_items;
    A.n_items = n_items;
    A.pixel_counter = ctx->d_pixel_counter;
    A.out = d_out;
    A.packed = packed;
    A.counters = ctx->d_counters;
    CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_pixel_counter, 0, sizeof(uint32_t), stream));
    if (o->collect_counters) CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, C_N * sizeof(unsigned long long), stream));
    if (n_items == 0) return CRT_OK;
    const uint32_t need_blocks = (n_items * 64u + BLOCK - 1) / BLOCK;
    const uint32_t blocks = need_blocks < ctx->grid_blocks ? need_blocks : ctx->grid_blocks;
    if (timed) CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, stream));
    if (o->collect_counters)
        hipLaunchKernelGGL(render_kernel<true>, dim3(blocks), dim3(BLOCK), 0, stream, A);
    else
        hipLaunchKernelGGL(render_kernel<false>, dim3(blocks), dim3(BLOCK), 0, stream, A);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    if (timed) CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, stream));
    return CRT_OK;
}

static int fetch_counters(crt_ctx *ctx, const crt_options *o, uint64_t pixels) {
    ctx->stats.pixels = pixels;
    ctx->stats.counters_valid = o->collect_counters ? 1 : 0;
    if (o->collect_counters) {
        unsigned long long c[C_N];
        CRT_HIP_CHECK(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
        ctx->stats.box_tests = c[C_BOX]; ctx->stats.tri_tests = c[C_TRI]; ctx->stats.leaf_index_reads = c[C_LEAFIDX];
        ctx->stats.shaded_hits = c[C_HIT]; ctx->stats.light_evals = c[C_LIGHT]; ctx->stats.texel_fetches = c[C_TEXEL];
        ctx->stats.primary_rays = c[C_PRIMARY]; ctx->stats.secondary_rays = c[C_SECONDARY]; ctx->stats.shadow_rays = c[C_SHADOW];
    }
    return CRT_OK;
}

extern "C" int crt_render(crt_ctx *ctx, const crt_options *o, const crt_rect *rects, uint32_t n_rects, float *out_rgb) {
    if (!ctx) return CRT_ERR_INVALID;
    int rc = check_options(ctx, o);
    if (rc) return rc;
    if (n_rects && !rects) { ctx->error = "rects is NULL"; return CRT_ERR_INVALID; }
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // coverage: the union of the clamped rectangles (RayTracer.cpp:84-85), as 8x8 tiles with lane masks
    bool same = ctx->cached_rects.size() == n_rects && n_rects > 0 &&
                memcmp(ctx->cached_rects.data(), rects, n_rects * sizeof(crt_rect)) == 0;
    if (!same) {
        const uint32_t tx = ctx->tiles_x, ty = ctx->tiles_y;
        std::vector<uint64_t> masks((size_t)tx * ty, 0);
        for (uint32_t r = 0; r < n_rects; r++) {
            uint64_t row_lim = (uint64_t)rects[r].row + rects[r].height, col_lim = (uint64_t)rects[r].col + rects[r].width;
            if (row_lim > ctx->height) row_lim = ctx->height;
            if (col_lim > ctx->width) col_lim = ctx->width;
            for (uint64_t row = rects[r].row; row < row_lim; row++) {
                for (uint64_t col = rects[r].col; col < col_lim;) {
                    uint64_t tcol = col / TILE, cend = (tcol + 1) * TILE;
                    if (cend > col_lim) cend = col_lim;
                    uint64_t bits = ((cend - col) >= 64 ? ~0ull : ((1ull << (cend - col)) - 1ull)) << ((row % TILE) * TILE + (col % TILE));
                    masks[(row / TILE) * tx + tcol] |= bits;
                    col = cend;
                }
            }
        }
        std::vector<WorkItem> items;
        uint64_t pixels = 0;
        for (uint32_t t = 0; t < tx * ty; t++)
            if (masks[t]) {
                items.push_back(WorkItem{t, t, masks[t]});
                pixels += (uint64_t)__builtin_popcountll(masks[t]);
            }
        rc = ensure_items(ctx, items.size() ? items.size() : 1);
        if (rc) return rc;
        if (!items.empty())
            CRT_HIP_CHECK(ctx, hipMemcpy(ctx->d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        ctx->cached_rects.assign(rects, rects + n_rects);
        ctx->cached_n_items = (uint32_t)items.size();
        ctx->cached_pixels = pixels;
    }
    hipEvent_t t0, t1;
    CRT_HIP_CHECK(ctx, hipEventCreate(&t0));
    CRT_HIP_CHECK(ctx, hipEventCreate(&t1));
    CRT_HIP_CHECK(ctx, hipEventRecord(t0, ctx->stream));
    rc = launch_render(ctx, o, ctx->cached_n_items, ctx->d_frame, 0, ctx->stream, true);
    if (rc) return rc;
    if (out_rgb)
        CRT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgb, ctx->d_frame, (size_t)ctx->width * ctx->height * 3 * sizeof(float),
                                          hipMemcpyDeviceToHost, ctx->stream));
    CRT_HIP_CHECK(ctx, hipEventRecord(t1, ctx->stream));
    CRT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    if (ctx->cached_n_items) {
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->stats.kernel_ms = ms;
    } else ctx->stats.kernel_ms = 0;
    CRT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, t0, t1));
    ctx->stats.total_ms = ms;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return fetch_counters(ctx, o, ctx->cached_pixels);
}

extern "C" uint32_t crt_packed_tile_count(const crt_ctx *ctx, uint32_t first, uint32_t stride) {
    if (!ctx || stride == 0) return 0;
    const uint32_t n = ctx->tiles_x * ctx->tiles_y;
    if (first >= n) return 0;
    return (n - first + stride - 1) / stride;
}

extern "C" int crt_render_tiles_device(crt_ctx *ctx, const crt_options *o, uint32_t first, uint32_t stride,
                                       float *d_packed, void *stream) {
    if (!ctx) return CRT_ERR_INVALID;
    int rc = check_options(ctx, o);
    if (rc) return rc;
    if (stride == 0 || !d_packed) { ctx->error = "bad tile partition"; return CRT_ERR_INVALID; }
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = crt_packed_tile_count(ctx, first, stride);
    // cached under a synthetic rect key {first, stride, ~0, ~0}
    crt_rect key{first, stride, 0xFFFFFFFFu, 0xFFFFFFFFu};
    bool same = ctx->cached_rects.size() == 1 && memcmp(ctx->cached_rects.data(), &key, sizeof(key)) == 0;
    if (!same) {
        std::vector<WorkItem> items(n);
        uint64_t pixels = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t t = first + j * stride;
            // lanes outside the image are masked in the kernel; count the real pixels here
            const uint32_t tx = t % ctx->tiles_x, ty = t / ctx->tiles_x;
            const uint32_t w = (tx + 1) * TILE <= ctx->width ? TILE : ctx->width - tx * TILE;
            const uint32_t h = (ty + 1) * TILE <= ctx->height ? TILE : ctx->height - ty * TILE;
            pixels += (uint64_t)w * h;
            items[j] = WorkItem{t, j, ~0ull};
        }
        rc = ensure_items(ctx, n ? n : 1);
        if (rc) return rc;
        if (n) CRT_HIP_CHECK(ctx, hipMemcpy(ctx->d_items, items.data(), n * sizeof(WorkItem), hipMemcpyHostToDevice));
        ctx->cached_rects.assign(1, key);
        ctx->cached_n_items = n;
        ctx->cached_pixels = pixels;
    }
    rc = launch_render(ctx, o, n, d_packed, 1, (hipStream_t)stream, true);
    if (rc) return rc;
    ctx->stats.pixels = ctx->cached_pixels;
    ctx->stats.counters_valid = 0;
    return CRT_OK;
}

extern "C" int crt_unpack_tiles_device(crt_ctx *ctx, const float *d_packed_all, uint32_t n_parts, uint64_t part_stride_floats,
                                       float *d_frame, void *stream) {
    if (!ctx || !d_packed_all || !d_frame || n_parts == 0) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n_tiles = ctx->tiles_x * ctx->tiles_y;
    const uint64_t threads = (uint64_t)n_tiles * 64;
    hipLaunchKernelGGL(unpack_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_packed_all,
                       n_parts, part_stride_floats, d_frame, ctx->width, ctx->height, ctx->tiles_x, n_tiles);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    return CRT_OK;
}

extern "C" int crt_quantize_device(crt_ctx *ctx, const float *d_rgb, uint64_t n_values, uint8_t *d_out, void *stream) {
    if (!ctx || !d_rgb || !d_out) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n_values == 0) return CRT_OK;
    hipLaunchKernelGGL(quantize_kernel, dim3((uint32_t)((n_values + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb,
                       n_values, d_out);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    return CRT_OK;
}

extern "C" int crt_read_quantized(crt_ctx *ctx, uint8_t *out_rgb8) {
    if (!ctx || !out_rgb8) return CRT_ERR_INVALID;
    const uint64_t n = (uint64_t)ctx->width * ctx->height * 3;
    int rc = crt_quantize_device(ctx, ctx->d_frame, n, ctx->d_quant, ctx->stream);
    if (rc) return rc;
    CRT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgb8, ctx->d_quant, n, hipMemcpyDeviceToHost, ctx->stream));
    CRT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CRT_OK;
}

extern "C" int crt_kernel_elapsed_ms(crt_ctx *ctx, double *ms) {
    if (!ctx || !ms) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    CRT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    float f = 0;
    CRT_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f;
    ctx->stats.kernel_ms = f;
    return CRT_OK;
}

extern "C" int crt_get_stats(crt_ctx *ctx, crt_stats *out) {
    if (!ctx || !out) return CRT_ERR_INVALID;
    *out = ctx->stats;
    return CRT_OK;
}

extern "C" int crt_synchronize(crt_ctx *ctx) {
    if (!ctx) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    CRT_HIP_CHECK(ctx, hipDeviceSynchronize());
    return CRT_OK;
}
