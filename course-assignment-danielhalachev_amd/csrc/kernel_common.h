// kernel_common.h -- device-side constants, argument block and exact-arithmetic helpers shared by all render kernels.
#pragma once

constexpr uint32_t END = CRT_LINK_END;
constexpr uint32_t LEAF = CRT_LINK_LEAF;
constexpr uint32_t LAST = CRT_ENTRY_LAST;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int TILE = 8;             // 8x8 pixel tiles: 64 pixels = one wavefront's worth
constexpr int FRAME_DWORDS = 8;     // per recursion level and lane
constexpr int FRAME_DWORDS_GI = 18; // ... in the GI mode (kernel_lane.h)
constexpr int BLOCK = 256;
constexpr float PI_F = 3.14159265358979323846f;  // M_PIf, RayTracer.cpp:27

enum : int { RAY_PRIMARY = 0, RAY_SHADOW = 1, RAY_REFLECTION = 2, RAY_REFRACTION = 3 };  // Ray.h:14
enum : int { ST_FETCH = 0, ST_TRAVERSE = 1, ST_DONE = 2 };
enum : int { FR_REFLECT = 0, FR_REFRACT_WAIT_REFLECTION = 1, FR_REFRACT_WAIT_REFRACTION = 2, FR_REFRACT_NO_TRANSMISSION = 3 };
enum : int { C_BOX = 0, C_TRI, C_LEAFIDX, C_HIT, C_LIGHT, C_TEXEL, C_PRIMARY, C_SECONDARY, C_SHADOW, C_N };  // crt_stats order
constexpr int C_PUBLIC = C_N;
constexpr int SC_OVERFLOW_WORD = 5 * 64 + 2;  // == SC_OVERFLOW of kernel_stream.h (static_assert there)

// ---- what the host side (crt_scene.hip, crt_launch.hip, crt_abi.hip) reads of the kernels' layouts ----
constexpr int MAX_GENERATIONS = 64;
// layout of KernelArgs::s_counts (uint32): [g] rays of level g, [SC_FETCH + g] fetch cursor of level g
//   [SC_HEAVY + g] rays evicted to the heavy kernel at level g, [SC_HEAVY_FETCH + g] / [SC_EVICT_FETCH + g] their cursors
enum : int { SC_COUNT = 0, SC_FETCH = MAX_GENERATIONS, SC_HEAVY = 2 * MAX_GENERATIONS, SC_HEAVY_FETCH = 3 * MAX_GENERATIONS,
             SC_EVICT_FETCH = 4 * MAX_GENERATIONS, SC_SHADOW = 5 * MAX_GENERATIONS, SC_SHADOW_FETCH, SC_OVERFLOW, SC_SHEAVY,
             SC_SHEAVY_FETCH, SC_SHEAVY_FETCH2, SC_GUARD, SC_SHADOW_SPLIT, SC_SHADOW_FETCH2, SC_SHEAVY_SPLIT,
             SC_WORDS,
             SC_HEAVY_DIAG = 384,  // diagnostics of a collect_counters == 2 render (kernel_heavy.h): 8 words closest-hit walks, 8 words shadow walks
             SC_ALLOC_WORDS = 512 };
// kernel_bvh.h, the level queue's words (FrameArgs::s_lq_words): rays reserved / claimed / finished, and a copy of the overflow word for
// the waves that wait -- 64 KB apart: hundreds of waves poll them, and words that share a memory channel share its request rate (with
// each other and with the bulk shadow pass's cursor, were they in the counter block)
enum : int { LQ_TAIL = 0, LQ_HEAD = 16384, LQ_DONE = 32768, LQ_ABORT = 49152, LQ_LEVEL0 = 57344, LQ_WORDS = 65536 };   // (LQ_LEVEL0: level 0 has ended -- nothing but the queue's own rays reserves entries any more)
static_assert(SC_WORDS <= SC_HEAVY_DIAG, "counter block too small");

static_assert(SC_OVERFLOW == SC_OVERFLOW_WORD, "kernel_common.h SC_OVERFLOW_WORD must match");
constexpr int PLAN_LEAF_DWORDS = 16;                        // kernel_plan.h: one top-level leaf of the plan
constexpr int PLAN_GROUP_LEAVES = 16, PLAN_GROUP_DWORDS = 8;  // ... and one group of the wide plan
constexpr uint32_t BVH_LDS_STACK = 16;                      // kernel_bvh.h: entries of a walk's stack that live in LDS

// constant address space: loads with a wave-uniform address become scalar loads (one s_load for the whole wave)
typedef const float __attribute__((address_space(4))) *kfp;
typedef const uint32_t __attribute__((address_space(4))) *ku32p;
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const v16f __attribute__((address_space(4))) *kv16p;

// A node is a leaf when bit 31 of its link is set -- except the all-ones END value, which an inner node without
// children (the root of an empty tree) carries as "nothing below, nothing after".
__host__ __device__ __forceinline__ bool is_leaf_link(uint32_t link) { return (link & LEAF) != 0 && link != END; }

struct DMaterial { float ax, ay, az, ior; uint32_t type, smooth; int32_t texture; uint32_t pad; };
struct DTexture { uint32_t kind; float ax, ay, az, bx, by, bz, scalar; uint32_t w, h; uint64_t offset; };

// Work item: one 8x8 tile, the lanes (pixels) of it that are to be rendered, and where its pixels go.
struct WorkItem { uint32_t tile; uint32_t out_tile; uint64_t mask; };

// Leaf sequence of one mesh tree for the wave-per-ray walk: level 0 = the leaves' own boxes in visit
// order, level k = union boxes of 64 consecutive entries of level k-1; the top level has <= 64 entries.
struct HeavyMesh {
    uint32_t n_levels;            // 0: the mesh has no leaves
    uint32_t first[4];            // first entry of each level in KernelArgs::hbox
    uint32_t count[4];            // entries per level
};

// What the kernels are given.  Three blocks, by how often they change (round 2 passed one 560 - 1000-byte struct by value:
// every field any code path touched was loaded at kernel entry and stayed in scalar registers, 60 - 130 of them spilled):
//   SceneArgs   written once by crt_create, resident in device memory: the flattened scene and everything derived from it
//   FrameArgs   one per frame, in a device slot of its own (frames are enqueued without waiting): camera, options, queues
//   KernelArgs  by value, 80 bytes: pointers to the two (constant address space: a field is one scalar load where it is used,
//               and a load on a cold path -- refill, eviction, shading -- is not hoisted into the walk loop) + what differs
//               between the launches of one frame
struct SceneArgs {
    const float4 *nodes;          // 2 x float4 per crt_node
    const uint32_t *leaf_tris;
    const uint32_t *leaf_meshes;
    const float4 *tris;           // 4 x float4 per crt_triangle
    const float4 *ltris;          // leaf-order copies: entry e of leaf_tris as {v0,nx} {v1,ny} {v2,nz} {plane, triangle id, last flag, -}
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
    uint32_t nested_boxes;        // every inner node's child boxes lie inside its own box (checked by crt_create)
    // heavy-ray path (kernel_heavy.h): leaf boxes in visit order + 64-ary group boxes above them
    const float4 *hbox;           // 2 x float4 per entry: {lo, begin|-} {hi, count|-}
    const HeavyMesh *hmesh;       // per mesh
    // single-leaf meshes (walls, floors: a root that is a leaf): the wave-per-ray kernels test them all in one step at the
    // start of a ray (kernel_heavy.h: TinyResults).  tiny_at[k] = the mesh's entry in hbox, tiny_flags[k] = crt_mesh::flags;
    // the device copy of crt_mesh::pad holds k + 1 for such a mesh, 0 for the others.
    // the top-level tree held in registers by the wave-per-ray kernels (kernel_heavy.h: TopRegs) when it is small:
    // its nodes are [top_first, top_first + top_count), top_count <= 64, leaf_meshes has <= 128 entries and there are <= 64 meshes
    uint32_t top_fast, top_first, top_count, top_leaf_entries, top_meshes;
    uint32_t plan_seq;            // plan_boxes holds the top-level tree's leaf sequence (any number of leaves): kernel_heavy.h walks it 64 leaves per instruction
    const uint32_t *tiny_at, *tiny_flags;
    uint32_t tiny_count;
    // The top-level tree as a PLAN for the per-lane kernels (kernel_plan.h), built by crt_create when the tree is small
    // (top_fast) and its boxes are nested: the top-level LEAVES in visit order.  A ray reaches a leaf exactly when the leaf's
    // own box passes (nesting + monotone slab test, as for the mesh trees), so a wave tests its rays against leaf k with k
    // uniform -- boxes in scalar registers, no gathers, no divergence -- instead of walking ~40 nodes and ~75 entries per lane.
    uint32_t plan_ok;             // the plan kernels may be used
    uint32_t plan_leaves;         // number of top-level leaves (plan_ok: <= 64)
    const float4 *plan_boxes;     // 4 x float4 per leaf: {lo, first entry in leaf_meshes} {hi, number of entries} {shadow mask words 0..3} {words 4..7}
    const float4 *plan_boxes_all; // the same leaves with masks over EVERY mesh (GI mode: shadow rays do not skip refractive meshes); order plan_shadow_mesh_all
    const uint32_t *plan_shadow_mesh_all;
    uint32_t plan_shadow_bits_all;
    const float4 *plan_groups;    // the wide plan: 2 x float4 per group of 16 consecutive leaves: {union lo, first leaf} {union hi, leaves}
    uint32_t plan_group_count;
    uint32_t plan_wide;           // the wide plan kernels may be used (more than 64 leaves or meshes, at most 256 meshes); plan_ok is then 0
    const uint32_t *plan_shadow_mesh;  // shadow order: bit b = mesh plan_shadow_mesh[b] (big meshes first: likeliest occluders)
    uint32_t plan_shadow_bits;    // number of non-refractive meshes with a bit in the shadow masks (<= 64; wide plan <= 256)
    // compact forms (crt_create): 3 x float4 per leaf entry {v0,nx} {v1,ny} {v2,nz}; the nodes again with leaf links
    // LEAF | (entries - 1) << 24 | first entry
    uint32_t plan_compact;
    const float4 *ptris;
    const float4 *pnodes;
    uint32_t plan_list_words;     // LDS words per lane of a closest-hit ray's mesh list: ceil(meshes / 4), at most 32 (wide plan: a longer list sends the ray to heavy_trace_closest)
    // The candidate filter (crt_bvh.h, kernel_bvh.h): a 4-wide hierarchy over the triangles grown by their acceptance margins, and the
    // inverse maps its hits are verified with against the reference's trees
    uint32_t bvh_ok;              // the filter exists (crt_create: every triangle's margin is bounded, the reference's boxes are nested)
    uint32_t bvh_stack;           // stack entries a walk can need: 3 x (inner nodes on the longest path) + 1
    float bvh_extent;             // largest absolute coordinate of any of its boxes
    float bvh_overlap_eps;        // slack of the box-overlap predicate of bvh_leaf_walk (kernel_bvh.h)
    const float4 *bvh_nodes;      // 8 x float4 per node: child boxes lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4], children[4], pad
    const float4 *bvh_vnodes;     // 6 x float4 per node: the children's reach boxes (crt_bvh.h: vnodes), for the miss check
    const float4 *bvh_cones;      // 4 x float4 per node: the children's normal cones (crt_bvh.h: cones)
    const float4 *bvh_tris;       // 3 x float4 per filter entry {v0,nx} {v1,ny} {v2,nz}, in the filter's leaf order
    const uint32_t *bvh_ids;      // per filter entry: triangle | BVH_ID_REFRACTIVE
    const uint32_t *tri_mesh;     // per triangle: its mesh | BVH_TRI_WALK (listed by too many leaves for a list: verified by the pruned tree walk)
    const uint32_t *tri_leaf_first;  // CSR per triangle: the reference leaves listing it, in visit order,
    const float4 *tri_leaf_list;     //   2 x float4 each: {box lo, entry position in leaf_triangles} {box hi, -}
    const uint32_t *mesh_top_first;  // CSR per mesh: the top-level leaves listing it, in visit order,
    const float4 *mesh_top_list;     //   the same form with the entry position in leaf_meshes
    // sizes of the arrays above, for the bounds-checked build of the filter kernels (crt_tuning::bvh = 2)
    uint32_t n_bvh_nodes, n_bvh_entries, n_triangles, n_nodes, n_leaf_tris, n_tri_leaf_entries, n_mesh_top_entries, n_meshes;
};

struct FrameArgs {
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
    // ray-stream buffers (kernel_stream.h)
    float4 *s_rayq[2];            // closest-hit ray queues of alternating recursion levels, 2 x float4 per ray
    float4 *s_shadowq;            // shadow rays of all levels: {origin, light distance}, {direction, light factor}
    uint8_t *s_occluded;          // one flag per shadow ray
    float4 *s_nodes;              // ray-tree nodes (TNode), 2 x float4 each
    uint32_t *s_counts;           // SC_WORDS counters / cursors, zeroed before every frame
    uint32_t s_ray_cap, s_shadow_cap, s_node_cap;
    uint32_t *fallback_total;     // frames redone by the queue-less kernel since crt_create (never reset)
    uint32_t *s_heavy;            // evicted ray ids of the current recursion level
    uint32_t *s_sheavy;           // evicted shadow ray ids (same capacity)
    uint32_t s_heavy_cap;
    float4 *s_hits;               // closest-hit records of evicted rays: {t, triangle, mesh, have}
    float4 *s_hits_all;           // ... of every ray of the current level, by ray index, when the per-lane walk leaves the shading to stream_shade_all
                                  //     (w: 0 / 1 no hit / hit, 2 evicted -- stream_shade_evicted shades it --, 3 not a ray)
    unsigned long long *s_lq;     // kernel_bvh.h / kernel_stream.h: the level queue, 8 granules per ray
    uint32_t s_lq_cap, lq_epoch;  // its capacity in rays; this frame's tag (never 0, never a tag the buffer may still hold)
    uint32_t *s_lq_words;         // its counters (kernel_stream.h: LQ_*), zeroed before every frame
    uint32_t *bvh_spill, *bvh_spill_side;  // kernel_bvh.h: what a walk's stack holds beyond its LDS part, one column per thread of the largest grid (level kernels / shadow passes)
    uint32_t heavy_level_threshold; // a recursion level with fewer rays than this goes to heavy_trace whole
    uint32_t fixed0;              // level 0's shadow rays go to fixed, tile-ordered slots (kernel_stream.h: level0_shadow_place)
    uint32_t use_gi, gi_samples, rays_per_pixel, gi_seed;  // crt_options: the GI / multi-sample mode (kernel_stream.h, kernel_lane.h, gi_random.h)
    uint32_t level0_samples;      // rays per pixel at level 0 of the ray-stream pass: max(1, rays_per_pixel) in the GI mode, else 1
    float monte_carlo_bias;
};

typedef const __attribute__((address_space(4))) SceneArgs *scene_args_p;
typedef const __attribute__((address_space(4))) FrameArgs *frame_args_p;

struct KernelArgs {
    scene_args_p s;
    frame_args_p f;
    unsigned long long *counters; // C_N, counting build only
    // crt_options::collect_counters == 2: the production kernels tally the box and triangle tests they EXECUTE (the exact
    // shortcuts make that fewer than the reference's, which the counting build tallies): {box tests, triangle tests}
    unsigned long long *exec_counters;
    unsigned long long *exec_plan;  // ... and the plan loops' box tests (wave-uniform boxes: no per-lane fetch), one word
    uint32_t exec_count;
    uint32_t step_budget;         // a lane's walk is evicted after this many steps (0 = never)
    uint32_t bundle;              // plan kernels: refill a wave when at most this many of its lanes are still walking (>= 64: at once)
    uint32_t only_if_overflow;    // lane kernel: run only when the stream pass overflowed its queues
    uint32_t force_whole;         // this level's per-lane kernel was not launched: the wave-per-ray kernel takes all its rays (kernel_stream.h)
    uint32_t wave_prio;           // s_setprio of the recursion levels' waves over the bulk shadow pass's, which shares the SIMDs with them
    uint32_t chunk;               // filter kernels: indices a wave claims from the launch's cursor per atomic (kernel_stream.h: wave_fetch_chunked)
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


// A ray as both kernels see it: origin, direction, 1/direction (BoundingBox.h:95 recomputes it per box,
// it only depends on the ray) and the mask of axes with |d| < FLT_EPSILON (BoundingBox.h:90).
struct Ray {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    uint32_t parmask;
};

__device__ __forceinline__ void ray_prepare(Ray &R) {
    R.parmask = (fabsf(R.dx) < FLT_EPSILON ? 1u : 0u) | (fabsf(R.dy) < FLT_EPSILON ? 2u : 0u) |
                (fabsf(R.dz) < FLT_EPSILON ? 4u : 0u);
    R.ix = 1.0f / R.dx;
    R.iy = 1.0f / R.dy;
    R.iz = 1.0f / R.dz;
}

// A mesh that overlaps several top-level leaves is listed in each of them, and the reference walks its tree again for
// every listed occurrence the ray reaches (KDTree.cpp:138-145, AccelerationStructure.cpp:62-77) -- on the benchmark scene
// two thirds of all box and triangle tests are such repeats.  A repeat finds the same hit as the first walk, and that hit
// can no longer change anything: the scene-level rule only takes a strictly smaller distance (KDTree.cpp:162; the first
// walk's own distance is not smaller than itself, and the running minimum only decreases), and a shadow ray's verdict is
// an OR over the walks.  So the production kernels walk every mesh once per ray, at its first occurrence in visit order,
// which is also where the reference first collects its hit.  (Meshes 128 and up are simply walked again; the counting
// build repeats everything, as its counters are the reference's.)
struct SeenMeshes { unsigned long long lo, hi; };   // meshes 0..127 already walked for this ray
__device__ __forceinline__ void seen_clear(SeenMeshes &seen) { seen.lo = 0; seen.hi = 0; }
__device__ __forceinline__ bool mesh_walk_is_repeat(SeenMeshes &seen, uint32_t mesh) {
    if (mesh >= 128u) return false;
    const unsigned long long bit = 1ull << (mesh & 63u);
    unsigned long long &word = mesh < 64u ? seen.lo : seen.hi;
    const bool repeat = (word & bit) != 0;
    word |= bit;
    return repeat;
}

// BoundingBox::hasIntersection (BoundingBox.h:85-108).  The reference returns early per axis; t0 only
// grows and t1 only shrinks and a NaN is never selected by std::max/std::min as written there, so
// (a) testing t0 > t1 once at the end gives the same verdict and (b) IEEE minNum/maxNum (v_min_f32 /
// v_max_f32) select the same values as the reference's compare-and-swap + std::max/std::min, up to the
// sign of a zero, which no comparison can see.  Axes with |d| < FLT_EPSILON take the containment test.
__device__ __forceinline__ bool slab_test_no_parallel(const Ray &R, float lox, float loy, float loz, float hix, float hiy,
                                                      float hiz) {
    const float ax = (lox - R.ox) * R.ix, bx = (hix - R.ox) * R.ix;
    const float ay = (loy - R.oy) * R.iy, by = (hiy - R.oy) * R.iy;
    const float az = (loz - R.oz) * R.iz, bz = (hiz - R.oz) * R.iz;
    const float t0 = fmaxf(fmaxf(fmaxf(-FLT_MAX, fminf(ax, bx)), fminf(ay, by)), fminf(az, bz));
    const float t1 = fminf(fminf(fminf(FLT_MAX, fmaxf(ax, bx)), fmaxf(ay, by)), fmaxf(az, bz));
    return !(t0 > t1);
}

__device__ __forceinline__ bool slab_test_general(const Ray &R, float lox, float loy, float loz, float hix, float hiy,
                                                  float hiz) {
    const float ax = (lox - R.ox) * R.ix, bx = (hix - R.ox) * R.ix;
    const float ay = (loy - R.oy) * R.iy, by = (hiy - R.oy) * R.iy;
    const float az = (loz - R.oz) * R.iz, bz = (hiz - R.oz) * R.iz;
    const bool px = R.parmask & 1u, py = R.parmask & 2u, pz = R.parmask & 4u;
    const bool reject = (px && ((R.ox < lox) || (R.ox > hix))) || (py && ((R.oy < loy) || (R.oy > hiy))) ||
                        (pz && ((R.oz < loz) || (R.oz > hiz)));
    // a parallel axis leaves t0 / t1 untouched (BoundingBox.h:90-93)
    const float nx = px ? -FLT_MAX : fminf(ax, bx), fx = px ? FLT_MAX : fmaxf(ax, bx);
    const float ny = py ? -FLT_MAX : fminf(ay, by), fy = py ? FLT_MAX : fmaxf(ay, by);
    const float nz = pz ? -FLT_MAX : fminf(az, bz), fz = pz ? FLT_MAX : fmaxf(az, bz);
    const float t0 = fmaxf(fmaxf(fmaxf(-FLT_MAX, nx), ny), nz);
    const float t1 = fminf(fminf(fminf(FLT_MAX, fx), fy), fz);
    return !(reject || (t0 > t1));
}

__device__ __forceinline__ bool slab_test(const Ray &R, float lox, float loy, float loz, float hix, float hiy, float hiz) {
    if (R.parmask == 0) return slab_test_no_parallel(R, lox, loy, loz, hix, hiy, hiz);
    return slab_test_general(R, lox, loy, loz, hix, hiy, hiz);
}

// Ray::intersectWithTriangle + Triangle::pointIsInTriangle (Ray.cpp:9-31, Triangle.cpp:37-57).
// a, b, c = vertex positions with the unit face normal in the .w lanes; plane = -(v0 . n) (Ray.cpp:17).
__device__ __forceinline__ bool triangle_test(const Ray &R, bool primary, const float4 &a, const float4 &b, const float4 &c,
                                              float plane, float &t_out) {
    const float nx = a.w, ny = b.w, nz = c.w;
    const float nd = dot3(R.dx, R.dy, R.dz, nx, ny, nz);
    if (primary && nd >= 0) return false;
    const float t = -(dot3(nx, ny, nz, R.ox, R.oy, R.oz) + plane) / nd;
    if (t < 0) return false;
    const float px = R.ox + R.dx * t, py = R.oy + R.dy * t, pz = R.oz + R.dz * t;
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

// RayTracer::getRay (RayTracer.cpp:61-80), pixel centre, followed by shootRay's own normalisation
// (RayTracer.cpp:420): every primary direction is normalised twice.
__device__ __forceinline__ void primary_ray(const KernelArgs &A, uint32_t px, uint32_t py, Ray &R) {
    float x = (float)px + 0.5f;
    float y = (float)py + 0.5f;
    x = x / (float)A.s->width;
    y = y / (float)A.s->height;
    x = (2.0f * x) - 1.0f;
    y = 1.0f - (2.0f * y);
    x = x * ((float)A.s->width / (float)A.s->height);
    const float z = -1.0f;
    R.dx = x * A.f->cam[0] + y * A.f->cam[3] + z * A.f->cam[6];   // row vector x matrix, Matrix.h:137-142
    R.dy = x * A.f->cam[1] + y * A.f->cam[4] + z * A.f->cam[7];
    R.dz = x * A.f->cam[2] + y * A.f->cam[5] + z * A.f->cam[8];
    normalize3(R.dx, R.dy, R.dz);
    R.ox = A.f->cam_pos[0]; R.oy = A.f->cam_pos[1]; R.oz = A.f->cam_pos[2];
    normalize3(R.dx, R.dy, R.dz);
    ray_prepare(R);
}

// float -> integer conversions as the reference's x86-64 build performs them (Texture.cpp:38-39, 67-68).  In range both
// targets truncate; out of range the GPU's conversions saturate, whereas cvttss2si returns the "integer indefinite"
// value -- and `(unsigned int)f` is compiled as a 64-bit cvttss2si whose low half is kept, so negative UVs wrap
// (-1.5 -> 0xFFFFFFFF) instead of clamping to 0.  Written out here so that the checker parity and the bitmap clamp see
// the same integers as the reference for every input, NaN included.
__device__ __forceinline__ int x86_float_to_int(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
}
__device__ __forceinline__ unsigned int x86_float_to_uint(float f) {
    const bool in_range = f >= -9223372036854775808.0f && f < 9223372036854775808.0f;  // NaN: false
    const long long v = in_range ? (long long)f : (long long)0x8000000000000000ull;
    return (unsigned int)(unsigned long long)v;
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
    const uint32_t i0 = A.s->tri_verts[3 * (size_t)tri], i1 = A.s->tri_verts[3 * (size_t)tri + 1],
                   i2 = A.s->tri_verts[3 * (size_t)tri + 2];
    // u * UV1 + v * UV2 + (w * UV0), Texture.cpp:34-36 / 63-65 (only x and y are used)
    const float uvx = (u * A.s->vuvs[3 * (size_t)i1] + v * A.s->vuvs[3 * (size_t)i2]) + w * A.s->vuvs[3 * (size_t)i0];
    const float uvy = (u * A.s->vuvs[3 * (size_t)i1 + 1] + v * A.s->vuvs[3 * (size_t)i2 + 1]) + w * A.s->vuvs[3 * (size_t)i0 + 1];
    if (T.kind == CRT_TEX_CHECKER) {
        const unsigned int x = x86_float_to_uint(uvx / T.scalar);
        const unsigned int y = x86_float_to_uint(uvy / T.scalar);
        if (x % 2 == y % 2) { r = T.ax; g = T.ay; b = T.az; } else { r = T.bx; g = T.by; b = T.bz; }
        return;
    }
    is_bitmap = true;
    int x = x86_float_to_int(uvx * (float)(int)T.w);
    int y = x86_float_to_int((1.0f - uvy) * (float)(int)T.h);
    x = (x < 0) ? 0 : (((int)T.w - 1 < x) ? (int)T.w - 1 : x);  // std::clamp
    y = (y < 0) ? 0 : (((int)T.h - 1 < y) ? (int)T.h - 1 : y);
    const uint32_t px = A.s->texels[T.offset + (size_t)y * T.w + (size_t)x];
    const float coefficient = 1.0f / 255.0f;  // Texture.cpp:53-57
    r = (float)(px & 255u) * coefficient;
    g = (float)((px >> 8) & 255u) * coefficient;
    b = (float)((px >> 16) & 255u) * coefficient;
}

// What shading needs to know about a closest hit (KDTree.cpp:168-190): point, (smooth) normal,
// barycentrics and the material of the mesh.
struct Surface {
    float px, py, pz, nx, ny, nz, u, v;
    DMaterial M;
};

__device__ __forceinline__ void surface_at(const KernelArgs &A, const Ray &R, float t, uint32_t tri, uint32_t mesh, Surface &S) {
    const float4 ta = A.s->tris[4 * (size_t)tri + 0], tb = A.s->tris[4 * (size_t)tri + 1], tc = A.s->tris[4 * (size_t)tri + 2];
    S.px = R.ox + R.dx * t; S.py = R.oy + R.dy * t; S.pz = R.oz + R.dz * t;  // Ray.cpp:23
    S.nx = ta.w; S.ny = tb.w; S.nz = tc.w;
    S.M = A.s->materials[A.s->meshes[mesh].material];
    S.u = 0; S.v = 0;
    if (S.M.smooth || S.M.texture >= 0) {
        // Triangle::getBarycentricCoordinates (Triangle.cpp:63-73)
        const float v0px = S.px - ta.x, v0py = S.py - ta.y, v0pz = S.pz - ta.z;
        const float e1x = tb.x - ta.x, e1y = tb.y - ta.y, e1z = tb.z - ta.z;
        const float e2x = tc.x - ta.x, e2y = tc.y - ta.y, e2z = tc.z - ta.z;
        const float area = len3(e1y * e2z - e1z * e2y, e1z * e2x - e1x * e2z, e1x * e2y - e1y * e2x);
        S.u = len3(v0py * e2z - v0pz * e2y, v0pz * e2x - v0px * e2z, v0px * e2y - v0py * e2x) / area;
        S.v = len3(e1y * v0pz - e1z * v0py, e1z * v0px - e1x * v0pz, e1x * v0py - e1y * v0px) / area;
        if (S.M.smooth) {  // KDTree.cpp:180-185: n1*u + n2*v + n0*(1-u-v), normalised
            const uint32_t i0 = A.s->tri_verts[3 * (size_t)tri], i1 = A.s->tri_verts[3 * (size_t)tri + 1],
                           i2 = A.s->tri_verts[3 * (size_t)tri + 2];
            const float w = 1 - S.u - S.v;
            S.nx = (A.s->vnormals[3 * (size_t)i1] * S.u + A.s->vnormals[3 * (size_t)i2] * S.v) + A.s->vnormals[3 * (size_t)i0] * w;
            S.ny = (A.s->vnormals[3 * (size_t)i1 + 1] * S.u + A.s->vnormals[3 * (size_t)i2 + 1] * S.v) + A.s->vnormals[3 * (size_t)i0 + 1] * w;
            S.nz = (A.s->vnormals[3 * (size_t)i1 + 2] * S.u + A.s->vnormals[3 * (size_t)i2 + 2] * S.v) + A.s->vnormals[3 * (size_t)i0 + 2] * w;
            normalize3(S.nx, S.ny, S.nz);
        }
    }
}

// One light of RayTracer::calculateDiffusion (RayTracer.cpp:308-318): the shadow ray towards it, its
// length and the factor (intensity / sphereArea * angle) the albedo is multiplied by when it is unoccluded.
__device__ __forceinline__ void light_setup(const KernelArgs &A, uint32_t li, float hpx, float hpy, float hpz, float hnx,
                                            float hny, float hnz, Ray &R, float &dist, float &kfac) {
    const float4 lg = A.s->lights[li];
    float lx = lg.x - hpx, ly = lg.y - hpy, lz = lg.z - hpz;
    dist = len3(lx, ly, lz);
    const float area = 4 * dist * dist * PI_F;
    normalize3(lx, ly, lz);
    const float angle = std_max(0.0f, dot3(lx, ly, lz, hnx, hny, hnz));
    kfac = lg.w / area * angle;
    R.ox = hpx + hnx * A.f->shadow_bias; R.oy = hpy + hny * A.f->shadow_bias; R.oz = hpz + hnz * A.f->shadow_bias;
    R.dx = lx; R.dy = ly; R.dz = lz;
    ray_prepare(R);
}
