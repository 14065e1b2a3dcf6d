// crt_device.hip -- the MI355X (gfx950) implementation behind include/crt_hip.h.
//
// What runs here is the reference's per-pixel hot path, tree mode, USE_GI = false
// (reference: SourceCode/src/RayTracer.cpp:61-112, 300-451, 507-517; src/KDTree.cpp:48-87,
// 127-192; src/AccelerationStructure.cpp:56-94; include/tracer/BoundingBox.h:85-108;
// src/Ray.cpp:9-31; src/Triangle.cpp:37-73; src/Texture.cpp:14-72; src/Color.cpp:12-16).
//
// Design (DESIGN.md has the measurements behind it):
//  * default path (kernel_stream.h): the reference's recursion unrolled BY RECURSION LEVEL.  Level g is one
//    launch of a per-lane walk kernel (kernel_plan.h: binary threaded nodes, the top-level tree as a plan) whose
//    lanes are refilled from a queue; it writes ray-tree nodes, child rays for level g+1 and shadow rays.  Walks that outlast a step budget, and whole small levels, go to a wave-per-ray kernel
//    (kernel_heavy.h).  The shadow rays of level 0 -- the bulk of the frame -- run on a side stream beside the
//    deeper levels.  stream_resolve evaluates every pixel's ray tree in the reference's post-order, so that
//    every float is combined in the same order;
//  * the reference's stack DFS has a FIXED visit order and no distance pruning, so the trees are flattened into
//    hit/miss links (crt_node) and walked without a stack; the plan and leaf-sequence forms rest on the boxes
//    being nested, which crt_create verifies;
//  * fallback, bit-exact and tested: render_lanes (kernel_lane.h: the full recursion per lane on an
//    explicit frame stack; redoes a frame whose queues overflowed, and renders the GI mode);
//  * arithmetic is IEEE binary32 with no contraction (-ffp-contract=off), correctly rounded divide and sqrt,
//    std::min/std::max semantics written out, so results are bit-identical to the x86-64 reference build.
//
// Kernel selection and sizing come in through crt_tuning (include/crt_hip.h); this file reads no environment
// variables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/crt_hip.h"
#include "crt_bvh.h"
#include "glibc_powf.h"

namespace {

#include "kernel_common.h"
#include "kernel_lane.h"
#include "kernel_stream.h"
#include "kernel_heavy.h"
#include "kernel_plan.h"
#include "kernel_bvh.h"

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

// crt_multi: scatter the parts' packed tiles into the frame; item j = {tile, slot in `packed`, mask of its rendered pixels}
__global__ void unpack_items_kernel(const float *packed, const WorkItem *items, uint32_t n_items, float *frame, uint32_t width,
                                    uint32_t height, uint32_t tiles_x) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (item, pixel)
    if (gid >= (uint64_t)n_items * 64) return;
    const WorkItem wi = items[gid >> 6];
    const uint32_t sub = (uint32_t)(gid & 63);
    const uint32_t px = (wi.tile % tiles_x) * TILE + (sub & 7u), py = (wi.tile / tiles_x) * TILE + (sub >> 3);
    if (!((wi.mask >> sub) & 1ull) || px >= width || py >= height) return;
    const float *src = packed + ((uint64_t)wi.out_tile * 64 + sub) * 3;
    float *dst = frame + ((uint64_t)py * width + px) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// unit-test kernel: the device build of csrc/glibc_powf.h
__global__ void pow5_kernel(const float *x, float *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = crt_pow5(x[i]);
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
    static constexpr int EV_RING = 64;       // event sets of the most recent render launches
    hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {}, ev2[EV_RING] = {}, ev3[EV_RING] = {}, ev4[EV_RING] = {};  // phase boundaries of a render
    hipEvent_t ev_fork[EV_RING] = {}, ev_s0[EV_RING] = {}, ev_s1[EV_RING] = {}, ev_s2[EV_RING] = {};        // ... of its side stream
    uint64_t launches = 0;
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0;
    SceneArgs scene{};                // host copy of what crt_create uploads to d_scene
    FrameArgs frame{};                // host copy of the NEXT frame's block (camera, queues ...)
    SceneArgs *d_scene = nullptr;
    FrameArgs *d_frame_ring = nullptr, *h_frame_ring = nullptr;  // one slot per launch in flight (EV_RING): device, and the pinned source of its copy
    std::vector<void *> allocs;
    float *d_frame = nullptr;
    uint8_t *d_quant = nullptr;
    WorkItem *d_items = nullptr;
    size_t items_cap = 0;
    std::vector<crt_rect> cached_rects;   // coverage the work items were built for: crt_render's rectangles ...
    bool cached_is_partition = false;     // ... or crt_render_tiles_device's {first, stride} (held in cached_rects[0].row / .col)
    uint32_t cached_n_items = 0;
    uint64_t cached_pixels = 0;
    uint32_t *d_sync = nullptr;      // [0] render_lanes' pixel counter
    uint64_t level_counters[C_N] = {}, shadow0_counters[C_N] = {};  // of the last counted render: recursion levels, bulk shadow pass
    enum Mode { MODE_STREAM, MODE_LANES } mode = MODE_STREAM;  // crt_tuning::mode (both produce identical pixels)
    float4 *d_rayq[2] = {nullptr, nullptr};
    float4 *d_shadowq = nullptr;
    uint8_t *d_occluded = nullptr;
    float4 *d_nodes = nullptr;
    uint32_t *d_scounts = nullptr;
    unsigned long long *d_exec = nullptr;           // executed-test tallies of a collect_counters == 2 render
    unsigned long long exec_counters[6] = {0, 0, 0, 0, 0, 0};  // {box, tri} x {all but shadow pass 0, shadow pass 0}, plan tests x the same
    uint32_t *d_fallback_total = nullptr;  // frames redone by the queue-less kernel since crt_create
    uint32_t *d_heavy = nullptr;      // evicted ray ids
    uint32_t *d_sheavy = nullptr;     // evicted shadow ray ids
    float4 *d_hits = nullptr;         // their closest hits
    float4 *d_hits_all = nullptr;     // a level's closest hits by ray index (kernel_plan.h: the walk-only builds)
    unsigned long long *d_lq = nullptr;   // kernel_bvh.h: the level queue (one entry per ray-tree node at most), 8 granules per ray
    uint32_t *d_lq_words = nullptr;       // ... and its counters (kernel_stream.h: LQ_*)
    uint32_t *d_bvh_spill = nullptr;  // kernel_bvh.h: the walks' stacks beyond their LDS part: one region for the caller's stream, one for the side stream
    hipStream_t side = nullptr;       // shadow pass 0 overlaps the deeper recursion levels on this stream
    // What a finished frame tells the next ones (queue sizing, launch sizes, fallback count): every frame copies its counter
    // block and the fallback total to ITS slot of this pinned ring, and the host reads a slot only once that frame's last
    // event has completed (harvest_counts), so launch decisions are a function of a completed frame, never of a copy in flight.
    static constexpr uint32_t H_SLOT_WORDS = 512 + 1;   // SC_ALLOC_WORDS + the fallback total
    uint32_t *h_ring = nullptr;       // EV_RING x H_SLOT_WORDS, pinned
    uint64_t next_count_harvest = 0;  // the oldest launch whose slot has not been read
    uint32_t slot_items[EV_RING] = {};               // work items of the frame in each slot
    std::vector<uint32_t> last_counts;               // the most recent COMPLETED frame's counter block (SC_* layout) ...
    uint32_t last_counts_items = 0;                  // ... and the work items it rendered (0: none yet)
    uint32_t fallbacks_seen = 0;                     // fallback total of that frame
    // queue sizing (ensure_stream): capacities as multiples of the frame's pixels, adapted from frame to frame
    double node_mult = 4.0, ray_mult = 2.0, shadow_extra = 1.0;  // (a GB at 1920x1080 with four lights: memory is not what this device lacks)
    uint32_t sizing_seen_fallbacks = 0;
    uint64_t queue_bytes = 0;         // bytes of the per-frame buffers as allocated now
    uint64_t regrows = 0;             // attempts repeated with larger queues (launch_render)
    uint32_t last_counts_cfg = 0;     // frame_config_of the frame last_counts came from
    uint32_t slot_cfg[64] = {};       // ... of the frame in each event slot
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr;  // around the last crt_render / crt_render_async call's device work
    bool pending = false;             // a frame enqueued by crt_render_async has not been waited for
    crt_options pending_options{};
    uint32_t heavy_cap = 0;
    uint32_t step_budget = 384;       // crt_tuning::step_budget (0 = never evict: also set when a mesh has too many leaves for the wave-per-ray walk)
    bool lean_ok = true;              // 32-bit byte offsets reach every node and leaf entry
    uint64_t stream_items = 0;        // work items the stream buffers are sized for
    uint64_t overflows = 0;           // frames redone by the fallback (diagnostic)
    uint32_t n_lights = 0;
    crt_tuning tuning{};
    std::string bvh_note;             // why this scene has no candidate filter (empty: it has one)
    std::string bvh_stats;            // ... and what it consists of
    unsigned long long *d_counters = nullptr;
    float *d_frames = nullptr;
    size_t frames_floats = 0;
    uint32_t grid_blocks = 0;
    crt_stats stats{};
    int num_cus = 0;
};

// constants that round 2 carried as crt_tuning fields (DESIGN.md section 7 has the measurements)
static constexpr uint32_t HEAVY_BLOCKS = 4096;   // grid of the wave-per-ray kernels (more blocks than fit: late ones balance the load)
static constexpr uint32_t REFILL_BUNDLE = 16;    // the plan kernels refill a wave when at most this many lanes still walk

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

// Unit-test hook: evaluates the device build of the restated glibc powf(x, 5) (csrc/glibc_powf.h) on host arrays.
extern "C" int crt_test_pow5(int device, const float *x, float *out, uint64_t n) {
    if (!x || !out) return CRT_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CRT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CRT_ERR_HIP;
    float *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&dx, n * sizeof(float)) != hipSuccess) return CRT_ERR_NOMEM;
    if (hipMalloc((void **)&dout, n * sizeof(float)) != hipSuccess) { (void)hipFree(dx); return CRT_ERR_NOMEM; }
    int rc = CRT_OK;
    if (hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = CRT_ERR_HIP;
    if (rc == CRT_OK && n) {
        hipLaunchKernelGGL(pow5_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, dx, dout, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_HIP;
    }
    (void)hipFree(dx);
    (void)hipFree(dout);
    return rc;
}

__host__ __device__ static inline uint32_t gi_test_value(uint32_t what, uint32_t a, uint32_t b) {
    float f;
    memcpy(&f, &a, sizeof(f));
    float r = 0;
    switch (what) {
        case 0: r = crt_sinf(f); break;
        case 1: r = crt_cosf(f); break;
        case 2: r = crt_gi_uniform(a, b); break;
        default: return crt_gi_mix(a, b);
    }
    uint32_t u;
    memcpy(&u, &r, sizeof(u));
    return u;
}
__global__ void gi_test_kernel(uint32_t what, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gi_test_value(what, a[i], b ? b[i] : 0u);
}
// Unit-test hook: the GI mode's sinf / cosf / generator (csrc/glibc_sincosf.h, csrc/gi_random.h) on host arrays.
extern "C" int crt_test_gi(int device, uint32_t what, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n) {
    if (!a || !out || what > 3u || (what >= 2u && !b)) return CRT_ERR_INVALID;
    if (device < 0) {
        for (uint64_t i = 0; i < n; i++) out[i] = gi_test_value(what, a[i], b ? b[i] : 0u);
        return CRT_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return CRT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CRT_ERR_HIP;
    uint32_t *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = CRT_OK;
    if (hipMalloc((void **)&da, (n ? n : 1) * 4) != hipSuccess || hipMalloc((void **)&db, (n ? n : 1) * 4) != hipSuccess ||
        hipMalloc((void **)&dout, (n ? n : 1) * 4) != hipSuccess) rc = CRT_ERR_NOMEM;
    if (rc == CRT_OK && n) {
        if (hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice) != hipSuccess || (b && hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice) != hipSuccess)) rc = CRT_ERR_HIP;
        if (rc == CRT_OK) {
            hipLaunchKernelGGL(gi_test_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, what, da, b ? db : nullptr, dout, n);
            if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_HIP;
        }
    }
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
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
    if (s->n_leaf_triangles && !s->leaf_triangles) return bad("missing leaf_triangles");
    if (s->n_leaf_meshes && !s->leaf_meshes) return bad("missing leaf_meshes");
    if (s->n_materials && !s->materials) return bad("missing materials");
    if (s->n_textures && !s->textures) return bad("missing textures");
    if (s->n_texels && !s->texels) return bad("missing texels");
    if (s->n_lights && !s->lights) return bad("missing lights");
    if (s->n_leaf_triangles > 0x7FFFFFFFull) return bad("too many leaf entries");
    // every index the kernel will follow is checked here, on the host, before anything is launched
    // Links must point FORWARD (nodes stored in visit order): this makes every walk finite whatever the
    // data, and it is what lets the packet kernel park a ray until `node index >= miss`.
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const crt_node &n = s->nodes[i];
        if (n.miss != CRT_LINK_END && (n.miss >= s->n_nodes || n.miss <= i)) return bad("node miss link must point forward");
        if (is_leaf_link(n.link)) continue;
        if (n.link != CRT_LINK_END && (n.link >= s->n_nodes || n.link <= i)) return bad("node hit link must point forward");
        if (n.link != CRT_LINK_END && n.miss != CRT_LINK_END && n.link > n.miss) return bad("node hit link beyond its miss link");
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
    // Leaf links must point inside THEIR entry array: a leaf of the top-level tree lists meshes, a leaf of a mesh tree
    // lists triangles.  Which tree a node belongs to is decided by reachability from top_root (links point forward,
    // so the walk below is finite); both arrays end with a terminated entry (checked above), so a leaf that begins
    // inside its array also ends inside it.
    {
        std::vector<bool> is_top(s->n_nodes, false);
        std::vector<uint32_t> stack{s->top_root};
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (i >= s->n_nodes || is_top[i]) continue;
            is_top[i] = true;
            const crt_node &n = s->nodes[i];
            if (is_leaf_link(n.link) || n.link == CRT_LINK_END) continue;
            stack.push_back(n.link);
            const uint32_t c2 = s->nodes[n.link].miss;
            if (c2 != n.miss && c2 != CRT_LINK_END) stack.push_back(c2);
        }
        for (uint32_t i = 0; i < s->n_nodes; i++) {
            const crt_node &n = s->nodes[i];
            if (!is_leaf_link(n.link)) continue;
            const uint64_t b = n.link & ~CRT_LINK_LEAF;
            if (is_top[i]) { if (b >= s->n_leaf_meshes) return bad("top-level leaf begins outside leaf_meshes"); }
            else if (b >= s->n_leaf_triangles) return bad("mesh-tree leaf begins outside leaf_triangles");
        }
        for (uint32_t i = 0; i < s->n_meshes; i++)
            if (is_top[s->meshes[i].root]) return bad("mesh root inside the top-level tree");
    }
    return CRT_OK;
}


extern "C" void crt_tuning_defaults(crt_tuning *t) {
    if (!t) return;
    memset(t, 0, sizeof(*t));
    t->size = (uint32_t)sizeof(*t);
    t->mode = CRT_MODE_STREAM;
    t->step_budget = 384; t->shadow_budget = 4096; t->level0_budget = 0;
    t->heavy_level = 100000; t->side_blocks = 2;
    t->node_cap = t->ray_cap = t->shadow_cap = 0;
    t->bvh = 1;
    t->level_queue = 2;
}

extern "C" int crt_create(const crt_scene_desc *s, int device, crt_ctx **out) { return crt_create_tuned(s, device, nullptr, out); }

extern "C" int crt_create_tuned(const crt_scene_desc *s, int device, const crt_tuning *tuning, crt_ctx **out) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    crt_tuning tune;
    crt_tuning_defaults(&tune);
    if (tuning) {
        if (tuning->size == 0 || tuning->size > sizeof(crt_tuning) || (tuning->size & 3u)) {
            g_create_error = "crt_tuning.size is not set (call crt_tuning_defaults first)";
            return CRT_ERR_INVALID;
        }
        memcpy(&tune, tuning, tuning->size);  // an older, shorter struct keeps the defaults of the newer fields
        tune.size = (uint32_t)sizeof(crt_tuning);
        if (tune.mode > CRT_MODE_LANES) { g_create_error = "crt_tuning.mode out of range"; return CRT_ERR_INVALID; }
    }
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
    {   // the side stream (bulk shadow pass) at the lowest stream priority: nothing waits for it until the levels are done
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CK(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, least));
    }
    for (int i = 0; i < crt_ctx::EV_RING; i++) {
        CK(hipEventCreate(&ctx->ev0[i]));
        CK(hipEventCreate(&ctx->ev1[i]));
        CK(hipEventCreate(&ctx->ev2[i]));
        CK(hipEventCreate(&ctx->ev3[i]));
        CK(hipEventCreate(&ctx->ev4[i]));
        CK(hipEventCreate(&ctx->ev_fork[i]));
        CK(hipEventCreate(&ctx->ev_s0[i]));
        CK(hipEventCreate(&ctx->ev_s1[i]));
        CK(hipEventCreate(&ctx->ev_s2[i]));
    }

    ctx->width = s->width;
    ctx->height = s->height;
    ctx->tiles_x = (s->width + TILE - 1) / TILE;
    ctx->tiles_y = (s->height + TILE - 1) / TILE;
    SceneArgs &A = ctx->scene;
    static_assert(sizeof(crt_node) == 32 && sizeof(crt_triangle) == 64, "record sizes");
    if (upload(ctx, (const float4 *)s->nodes, (size_t)s->n_nodes * 2, &A.nodes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_triangles, (size_t)s->n_leaf_triangles, &A.leaf_tris)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_meshes, (size_t)s->n_leaf_meshes, &A.leaf_meshes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, (const float4 *)s->triangles, (size_t)s->n_triangles * 4, &A.tris)) return fail(CRT_ERR_HIP);
    {
        // leaf-order triangle records: what a walk reads inside a leaf, without the index indirection
        std::vector<float4> lt((size_t)s->n_leaf_triangles * 4);
        for (uint64_t e = 0; e < s->n_leaf_triangles; e++) {
            const uint32_t ent = s->leaf_triangles[e], tri = ent & ~CRT_ENTRY_LAST;
            const crt_triangle &T = s->triangles[tri];
            lt[4 * e + 0] = make_float4(T.v0[0], T.v0[1], T.v0[2], T.nx);
            lt[4 * e + 1] = make_float4(T.v1[0], T.v1[1], T.v1[2], T.ny);
            lt[4 * e + 2] = make_float4(T.v2[0], T.v2[1], T.v2[2], T.nz);
            float idbits, lastbits;
            const uint32_t last = (ent & CRT_ENTRY_LAST) ? 1u : 0u;
            memcpy(&idbits, &tri, 4);
            memcpy(&lastbits, &last, 4);
            lt[4 * e + 3] = make_float4(T.plane, idbits, lastbits, 0.0f);
        }
        if (upload(ctx, lt.data(), lt.size(), &A.ltris)) return fail(CRT_ERR_HIP);
    }
    // Compact forms for the plan kernels (kernel_plan.h), whose cost is the number of vector-memory instructions per step:
    // a triangle as 3 x float4 {v0,nx} {v1,ny} {v2,nz} (the plane offset is recomputed, -(v0 . n) as on the host, Ray.cpp:17;
    // the triangle's id is looked up in leaf_tris only for a hit) and a leaf node's link as LEAF | (count - 1) << 24 | first
    // entry, so that a walk knows where its leaf ends without the record's `last` flag.
    std::vector<uint32_t> compact_link(s->n_nodes);
    {
        bool ok = s->n_leaf_triangles < (1ull << 24);
        std::vector<float4> pn((size_t)s->n_nodes * 2);
        for (uint32_t i = 0; i < s->n_nodes; i++) {
            const crt_node &n = s->nodes[i];
            uint32_t link = n.link;
            if (is_leaf_link(n.link)) {
                const uint64_t begin = n.link & ~CRT_LINK_LEAF;
                uint64_t count = 0;
                if (begin < s->n_leaf_triangles) {  // (a top-level leaf's link indexes leaf_meshes: whatever this gives is not used)
                    uint64_t e = begin;
                    do { count++; } while (!(s->leaf_triangles[e++] & CRT_ENTRY_LAST) && e < s->n_leaf_triangles);
                }
                if (count >= 1 && count <= 128 && begin < (1ull << 24)) link = CRT_LINK_LEAF | (uint32_t)((count - 1) << 24) | (uint32_t)begin;
                else if (begin < s->n_leaf_triangles) ok = false;
            }
            compact_link[i] = link;
            float lb, mb;
            memcpy(&lb, &link, 4);
            memcpy(&mb, &n.miss, 4);
            pn[2 * (size_t)i] = make_float4(n.lo[0], n.lo[1], n.lo[2], mb);
            pn[2 * (size_t)i + 1] = make_float4(n.hi[0], n.hi[1], n.hi[2], lb);
        }
        std::vector<float4> pt(ok ? (size_t)s->n_leaf_triangles * 3 : 0);
        for (uint64_t e = 0; ok && e < s->n_leaf_triangles; e++) {
            const crt_triangle &T = s->triangles[s->leaf_triangles[e] & ~CRT_ENTRY_LAST];
            pt[3 * e + 0] = make_float4(T.v0[0], T.v0[1], T.v0[2], T.nx);
            pt[3 * e + 1] = make_float4(T.v1[0], T.v1[1], T.v1[2], T.ny);
            pt[3 * e + 2] = make_float4(T.v2[0], T.v2[1], T.v2[2], T.nz);
            // the kernels recompute the plane offset: it must be the stored one, bit for bit, or this form is not used
            const float plane = -(T.v0[0] * T.nx + T.v0[1] * T.ny + T.v0[2] * T.nz);
            if (memcmp(&plane, &T.plane, 4) != 0) ok = false;
        }
        if (!ok) { pt.clear(); pn.clear(); }
        A.plan_compact = ok ? 1u : 0u;
        if (upload(ctx, pt.data(), pt.size(), &A.ptris)) return fail(CRT_ERR_HIP);
        if (upload(ctx, pn.data(), pn.size(), &A.pnodes)) return fail(CRT_ERR_HIP);
    }
    std::vector<HeavyMesh> hmesh_host;  // filled with the leaf sequences below, read again for the single-leaf mesh table
    std::vector<bool> is_top(s->n_nodes, false);
    bool top_is_range = false;  // the top-level tree's nodes are ONE index range [top_first, top_first + top_count)
    {
        // the top-level tree's nodes: reachable from top_root (links point forward, so the walk is finite)
        std::vector<uint32_t> stack{s->top_root};
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (i >= s->n_nodes || is_top[i]) continue;
            is_top[i] = true;
            const crt_node &n = s->nodes[i];
            if (is_leaf_link(n.link) || n.link == CRT_LINK_END) continue;
            stack.push_back(n.link);
            const uint32_t c2 = s->nodes[n.link].miss;
            if (c2 != n.miss && c2 != CRT_LINK_END) stack.push_back(c2);
        }
        uint32_t lo = UINT32_MAX, hi = 0, cnt = 0;
        for (uint32_t i = 0; i < s->n_nodes; i++)
            if (is_top[i]) { lo = i < lo ? i : lo; hi = i > hi ? i : hi; cnt++; }
        top_is_range = cnt > 0 && hi - lo + 1 == cnt;
        A.top_fast = (top_is_range && cnt <= 64u && s->n_leaf_meshes <= 128u && s->n_meshes <= 64u) ? 1u : 0u;
        A.top_first = cnt ? lo : 0u;
        A.top_count = cnt;
        A.top_leaf_entries = s->n_leaf_meshes;
        A.top_meshes = s->n_meshes;
    }
    {
        // Leaf sequence of every mesh tree (kernel_heavy.h): the leaves' own boxes in visit order, then union
        // boxes of 64 entries per level until at most 64 remain.  With forward links the nodes of a mesh tree
        // are the index range [root, next tree's root), already in visit order.
        std::vector<uint32_t> roots;
        roots.push_back(s->top_root);
        for (uint32_t m = 0; m < s->n_meshes; m++) roots.push_back(s->meshes[m].root);
        std::sort(roots.begin(), roots.end());
        std::vector<float4> hbox;
        std::vector<HeavyMesh> &hm = hmesh_host;
        hm.assign(s->n_meshes, HeavyMesh{});
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            HeavyMesh &H = hm[m];
            memset(&H, 0, sizeof(H));
            const uint32_t root = s->meshes[m].root;
            auto it = std::upper_bound(roots.begin(), roots.end(), root);
            const uint32_t end = it == roots.end() ? s->n_nodes : *it;
            std::vector<float4> level;  // 2 x float4 per entry
            for (uint32_t i = root; i < end; i++) {
                const crt_node &n = s->nodes[i];
                if (!is_leaf_link(n.link)) continue;
                const uint32_t begin = n.link & ~CRT_LINK_LEAF;
                uint32_t count = 0;
                if (begin < s->n_leaf_triangles) {
                    uint64_t e = begin;
                    do { count++; } while (!(s->leaf_triangles[e++] & CRT_ENTRY_LAST) && e < s->n_leaf_triangles);
                }
                float bb, cb;
                memcpy(&bb, &begin, 4);
                memcpy(&cb, &count, 4);
                level.push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], bb));
                level.push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], cb));
            }
            uint32_t nl = 0;
            while (!level.empty() && nl < 4) {
                const uint32_t cnt = (uint32_t)(level.size() / 2);
                H.first[nl] = (uint32_t)(hbox.size() / 2);
                H.count[nl] = cnt;
                hbox.insert(hbox.end(), level.begin(), level.end());
                nl++;
                if (cnt <= 64) break;
                std::vector<float4> up;
                for (uint32_t g = 0; g < cnt; g += 64) {
                    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    for (uint32_t k = g; k < cnt && k < g + 64; k++) {
                        const float4 &a = level[2 * k], &b = level[2 * k + 1];
                        lo[0] = a.x < lo[0] ? a.x : lo[0]; lo[1] = a.y < lo[1] ? a.y : lo[1]; lo[2] = a.z < lo[2] ? a.z : lo[2];
                        hi[0] = b.x > hi[0] ? b.x : hi[0]; hi[1] = b.y > hi[1] ? b.y : hi[1]; hi[2] = b.z > hi[2] ? b.z : hi[2];
                    }
                    up.push_back(make_float4(lo[0], lo[1], lo[2], 0.0f));
                    up.push_back(make_float4(hi[0], hi[1], hi[2], 0.0f));
                }
                level.swap(up);
            }
            // more than 64^4 leaves: leave n_levels = 0 for this mesh -> the heavy path is switched off below
            H.n_levels = (!level.empty() && H.count[nl ? nl - 1 : 0] <= 64) ? nl : 0;
            if (H.n_levels == 0 && !level.empty()) ctx->step_budget = 0;
        }
        if (upload(ctx, hbox.data(), hbox.size(), &A.hbox)) return fail(CRT_ERR_HIP);
        if (upload(ctx, hm.data(), hm.size(), &A.hmesh)) return fail(CRT_ERR_HIP);
    }
    if (upload(ctx, s->triangle_vertices, (size_t)s->n_triangles * 3, &A.tri_verts)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->vertex_normals, (size_t)s->n_vertices * 3, &A.vnormals)) return fail(CRT_ERR_HIP);
    if (s->vertex_uvs) {
        if (upload(ctx, s->vertex_uvs, (size_t)s->n_vertices * 3, &A.vuvs)) return fail(CRT_ERR_HIP);
    } else A.vuvs = nullptr;
    {
        // the meshes' device copy marks single-leaf meshes (kernel_heavy.h: TinyResults): at most 64 of them, at most 512 triangles in all
        std::vector<crt_mesh> dm(s->meshes, s->meshes + s->n_meshes);
        std::vector<uint32_t> tiny_at, tiny_flags;
        uint64_t tiny_tris = 0;
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            dm[m].pad = 0;
            const crt_node &root = s->nodes[s->meshes[m].root];
            const HeavyMesh &H = hmesh_host[m];
            if (!is_leaf_link(root.link) || H.n_levels != 1 || H.count[0] != 1 || tiny_at.size() >= 64) continue;
            uint32_t cnt = 0;
            for (uint64_t e = root.link & ~CRT_LINK_LEAF; e < s->n_leaf_triangles; e++) { cnt++; if (s->leaf_triangles[e] & CRT_ENTRY_LAST) break; }
            if (tiny_tris + cnt > 512) continue;
            tiny_tris += cnt;
            dm[m].pad = (uint32_t)tiny_at.size() + 1u;
            tiny_at.push_back(H.first[0]);
            tiny_flags.push_back(s->meshes[m].flags);
        }
        A.tiny_count = (uint32_t)tiny_at.size();
        if (upload(ctx, tiny_at.data(), tiny_at.size(), &A.tiny_at)) return fail(CRT_ERR_HIP);
        if (upload(ctx, tiny_flags.data(), tiny_flags.size(), &A.tiny_flags)) return fail(CRT_ERR_HIP);
        if (upload(ctx, dm.data(), dm.size(), &A.meshes)) return fail(CRT_ERR_HIP);
    }
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
    {
        // Are child boxes nested in their parent's box?  (True for every tree the reference builds.)  With
        // forward links, the nodes in (i, miss_i) are exactly the descendants of inner node i.
        bool nested = true;
        std::vector<uint32_t> stack;  // enclosing inner nodes of the current position
        std::vector<bool> root(s->n_nodes, false);
        root[s->top_root] = true;
        for (uint32_t m = 0; m < s->n_meshes; m++) root[s->meshes[m].root] = true;
        for (uint32_t i = 0; i < s->n_nodes && nested; i++) {
            while (!stack.empty() && s->nodes[stack.back()].miss != CRT_LINK_END && s->nodes[stack.back()].miss <= i) stack.pop_back();
            // a tree root starts a new nesting chain: it is not a descendant of the previous tree's nodes
            if (root[i]) stack.clear();
            if (!stack.empty()) {
                const crt_node &p = s->nodes[stack.back()], &c = s->nodes[i];
                for (int a = 0; a < 3; a++)
                    if (!(c.lo[a] >= p.lo[a] && c.hi[a] <= p.hi[a])) nested = false;
            }
            if (!is_leaf_link(s->nodes[i].link)) stack.push_back(i);
        }
        A.nested_boxes = nested ? 1u : 0u;
    }
    {
        // The plan of the top-level tree (kernel_plan.h; kernel_heavy.h walks the same table as a leaf sequence): its leaves in visit order, which is index order.
        std::vector<float4> boxes, boxes_all;
        std::vector<uint32_t> order, order_all;  // non-refractive meshes / every mesh (the GI mode's shadow rays), most leaves first
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            if (!(s->meshes[m].flags & 1u)) order.push_back(m);
            order_all.push_back(m);
        }
        auto more_leaves = [&](uint32_t a, uint32_t b) { return hmesh_host[a].count[0] > hmesh_host[b].count[0]; };
        std::stable_sort(order.begin(), order.end(), more_leaves);
        std::stable_sort(order_all.begin(), order_all.end(), more_leaves);
        constexpr uint32_t MAX_BITS = 256;  // shadow mask bits of the wide plan (kernel_plan.h: 8 words); the plan proper uses the first 64
        std::vector<uint32_t> bit_of(s->n_meshes, MAX_BITS), bit_of_all(s->n_meshes, MAX_BITS);
        for (size_t b = 0; b < order.size() && b < MAX_BITS; b++) bit_of[order[b]] = (uint32_t)b;
        for (size_t b = 0; b < order_all.size() && b < MAX_BITS; b++) bit_of_all[order_all[b]] = (uint32_t)b;
        // the plan reads the top-level leaves off the index range [top_first, top_first + top_count): a scene whose top-level nodes are
        // interleaved with mesh-tree nodes (forward links allow it) has no plan and stays on the faithful kernels
        const bool contiguous = top_is_range && A.nested_boxes;
        uint32_t n_leaves = 0;
        std::vector<float4> groups;
        for (uint32_t i = A.top_first; contiguous && i < A.top_first + A.top_count; i++) {
            const crt_node &n = s->nodes[i];
            if (!is_leaf_link(n.link)) continue;
            const uint32_t begin = n.link & ~CRT_LINK_LEAF;
            uint32_t count = 0;
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mask_all[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t e = begin; e < s->n_leaf_meshes; e++) {
                const uint32_t mi = s->leaf_meshes[e] & ~CRT_ENTRY_LAST;
                if (bit_of[mi] < MAX_BITS) mask[bit_of[mi] >> 5] |= 1u << (bit_of[mi] & 31u);
                if (bit_of_all[mi] < MAX_BITS) mask_all[bit_of_all[mi] >> 5] |= 1u << (bit_of_all[mi] & 31u);
                count++;
                if (s->leaf_meshes[e] & CRT_ENTRY_LAST) break;
            }
            float bb, cb, mf[8];
            memcpy(&bb, &begin, 4);
            memcpy(&cb, &count, 4);
            for (std::vector<float4> *table : {&boxes, &boxes_all}) {
                memcpy(mf, table == &boxes ? mask : mask_all, sizeof(mf));
                table->push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], bb));  // PLAN_LEAF_DWORDS = 16 per leaf (kernel_plan.h)
                table->push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], cb));
                table->push_back(make_float4(mf[0], mf[1], mf[2], mf[3]));
                table->push_back(make_float4(mf[4], mf[5], mf[6], mf[7]));
            }
            // the wide plan's groups: PLAN_GROUP_LEAVES consecutive leaves under their union box
            if (n_leaves % PLAN_GROUP_LEAVES == 0) {
                float first_leaf, zero = 0.0f;
                memcpy(&first_leaf, &n_leaves, 4);
                groups.push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], first_leaf));
                groups.push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], zero));
            }
            float4 &glo = groups[groups.size() - 2], &ghi = groups[groups.size() - 1];
            glo.x = std::min(glo.x, n.lo[0]); glo.y = std::min(glo.y, n.lo[1]); glo.z = std::min(glo.z, n.lo[2]);
            ghi.x = std::max(ghi.x, n.hi[0]); ghi.y = std::max(ghi.y, n.hi[1]); ghi.z = std::max(ghi.z, n.hi[2]);
            const uint32_t in_group = n_leaves % PLAN_GROUP_LEAVES + 1u;
            memcpy(&ghi.w, &in_group, 4);
            n_leaves++;
        }
        // the per-lane plan kernels keep a ray's meshes in two 32-bit words (the plan proper) or eight (the wide plan)
        A.plan_ok = (contiguous && A.top_fast && n_leaves <= 64u && s->n_meshes <= 64u && A.plan_compact) ? 1u : 0u;
        A.plan_wide = (!A.plan_ok && contiguous && n_leaves > 0 && s->n_meshes <= MAX_BITS && A.plan_compact) ? 1u : 0u;
        A.plan_leaves = n_leaves;
        A.plan_seq = (contiguous && n_leaves > 0) ? 1u : 0u;  // the wave-per-ray kernels walk the leaf sequence (kernel_heavy.h)
        A.plan_group_count = (uint32_t)(groups.size() / 2);
        A.plan_shadow_bits = (uint32_t)std::min<size_t>(order.size(), A.plan_ok ? 64u : MAX_BITS);
        A.plan_list_words = std::min((s->n_meshes + 3u) / 4u, 32u);
        if (upload(ctx, groups.data(), groups.size(), &A.plan_groups)) return fail(CRT_ERR_HIP);
        if (upload(ctx, boxes.data(), boxes.size(), &A.plan_boxes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, order.data(), order.size(), &A.plan_shadow_mesh)) return fail(CRT_ERR_HIP);
        A.plan_shadow_bits_all = (uint32_t)std::min<size_t>(order_all.size(), A.plan_ok ? 64u : MAX_BITS);
        if (upload(ctx, boxes_all.data(), boxes_all.size(), &A.plan_boxes_all)) return fail(CRT_ERR_HIP);
        if (upload(ctx, order_all.data(), order_all.size(), &A.plan_shadow_mesh_all)) return fail(CRT_ERR_HIP);
    }
    {
        // The candidate filter (crt_bvh.h): built on the host, once; a scene it cannot be built for renders on the reference-order kernels
        BvhHost bvh;
        bvh_build(s, A.nested_boxes != 0, bvh);
        if (bvh.ok && !A.plan_compact) { bvh.ok = false; bvh.why = "no compact leaf links"; }  // (kernel_bvh.h: bvh_leaf_walk reads them)
        A.bvh_stack = 3u * bvh.wide_depth + 1u;   // what a walk can push: three children per inner node of a path
        if (bvh.ok && A.bvh_stack > 1024u) { bvh.ok = false; bvh.why = "hierarchy too deep"; }
        ctx->bvh_note = bvh.ok ? "" : bvh.why;
        if (bvh.ok) {
            char note[160];
            snprintf(note, sizeof(note), "nodes:%zu,entries:%zu,depth:%u/%u,walk_triangles:%u,max_margin:%.3g", bvh.nodes.size(), bvh.ids.size(),
                     bvh.max_depth, bvh.wide_depth, bvh.walk_triangles, bvh.max_margin);
            ctx->bvh_stats = note;
        }
        // the leaf sequences of the wave-per-ray kernels were read off index ranges [root, next root): a description whose trees are
        // interleaved (forward links allow it) is walked by the faithful kernels, which only follow links
        if (!bvh.trees_are_ranges) ctx->step_budget = 0;
        A.bvh_ok = bvh.ok ? 1u : 0u;
        A.bvh_extent = bvh.extent;
        A.bvh_overlap_eps = bvh.overlap_eps;
        A.n_bvh_nodes = (uint32_t)bvh.nodes.size(); A.n_bvh_entries = (uint32_t)bvh.ids.size(); A.n_triangles = s->n_triangles; A.n_nodes = s->n_nodes;
        A.n_leaf_tris = (uint32_t)s->n_leaf_triangles; A.n_tri_leaf_entries = (uint32_t)(bvh.tri_leaf_list.size() / 8);
        A.n_mesh_top_entries = (uint32_t)(bvh.mesh_top_list.size() / 8); A.n_meshes = s->n_meshes;
        if (!bvh.ok) bvh = BvhHost{};
        static_assert(sizeof(BvhNode) == 8 * sizeof(float4), "BvhNode = 8 x float4");
        if (upload(ctx, (const float4 *)bvh.nodes.data(), bvh.nodes.size() * 8, &A.bvh_nodes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.vnodes.data(), bvh.vnodes.size() / 4, &A.bvh_vnodes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.cones.data(), bvh.cones.size() / 4, &A.bvh_cones)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.tris.data(), bvh.tris.size() / 4, &A.bvh_tris)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.ids.data(), bvh.ids.size(), &A.bvh_ids)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.tri_mesh.data(), bvh.tri_mesh.size(), &A.tri_mesh)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.tri_leaf_first.data(), bvh.tri_leaf_first.size(), &A.tri_leaf_first)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.tri_leaf_list.data(), bvh.tri_leaf_list.size() / 4, &A.tri_leaf_list)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.mesh_top_first.data(), bvh.mesh_top_first.size(), &A.mesh_top_first)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.mesh_top_list.data(), bvh.mesh_top_list.size() / 4, &A.mesh_top_list)) return fail(CRT_ERR_HIP);
    }
    A.bgx = s->background[0]; A.bgy = s->background[1]; A.bgz = s->background[2];
    A.width = s->width; A.height = s->height; A.tiles_x = ctx->tiles_x;
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(ctx->frame.cam, ident, sizeof(ident));
    ctx->frame.cam_pos[0] = ctx->frame.cam_pos[1] = ctx->frame.cam_pos[2] = 0;

    size_t frame_bytes = (size_t)s->width * s->height * 3 * sizeof(float);
    CK(hipMalloc((void **)&ctx->d_frame, frame_bytes));
    CK(hipMemset(ctx->d_frame, 0, frame_bytes));  // colorBuffer starts as Color() = (0,0,0), RayTracer.cpp:46-50
    CK(hipMalloc((void **)&ctx->d_quant, (size_t)s->width * s->height * 3));
    CK(hipMalloc((void **)&ctx->d_sync, 4 * sizeof(uint32_t)));
    ctx->tuning = tune;
    ctx->mode = tune.mode == CRT_MODE_LANES ? crt_ctx::MODE_LANES : crt_ctx::MODE_STREAM;
    if (ctx->step_budget) ctx->step_budget = tune.step_budget;  // (0: a mesh with too many leaves switched the wave-per-ray path off)
    ctx->lean_ok = s->n_nodes < (1u << 27) && s->n_leaf_triangles < (1ull << 26);
    CK(hipMalloc((void **)&ctx->d_lq_words, LQ_WORDS * sizeof(uint32_t)));
    CK(hipMemset(ctx->d_lq_words, 0, LQ_WORDS * sizeof(uint32_t)));
    CK(hipMalloc((void **)&ctx->d_exec, 6 * sizeof(unsigned long long)));
    CK(hipMemset(ctx->d_exec, 0, 6 * sizeof(unsigned long long)));
    CK(hipMalloc((void **)&ctx->d_scounts, SC_ALLOC_WORDS * sizeof(uint32_t)));
    CK(hipMemset(ctx->d_scounts, 0, SC_ALLOC_WORDS * sizeof(uint32_t)));
    static_assert(crt_ctx::H_SLOT_WORDS == SC_ALLOC_WORDS + 1, "pinned slot = counter block + fallback total");
    CK(hipHostMalloc((void **)&ctx->h_ring, (size_t)crt_ctx::EV_RING * crt_ctx::H_SLOT_WORDS * sizeof(uint32_t)));
    memset(ctx->h_ring, 0, (size_t)crt_ctx::EV_RING * crt_ctx::H_SLOT_WORDS * sizeof(uint32_t));
    ctx->last_counts.assign(SC_ALLOC_WORDS, 0u);
    CK(hipMalloc((void **)&ctx->d_fallback_total, sizeof(uint32_t)));
    CK(hipMemset(ctx->d_fallback_total, 0, sizeof(uint32_t)));
    ctx->n_lights = s->n_lights;
    CK(hipMalloc((void **)&ctx->d_counters, 3 * C_N * sizeof(unsigned long long)));  // [levels | shadow pass 0 | the rest]
    // persistent grid: 8 blocks of 256 threads per CU gives every CU its 32 waves if registers allow
    ctx->grid_blocks = (uint32_t)ctx->num_cus * 8u;
    if (ctx->scene.bvh_ok && ctx->scene.bvh_stack > BVH_LDS_STACK)
        CK(hipMalloc((void **)&ctx->d_bvh_spill, 2u * (size_t)ctx->grid_blocks * BLOCK * (ctx->scene.bvh_stack - BVH_LDS_STACK) * sizeof(uint32_t)));
    // the argument blocks (kernel_common.h): the scene's once, a slot per frame in flight for the frames'
    CK(hipMalloc((void **)&ctx->d_scene, sizeof(SceneArgs)));
    CK(hipMemcpy(ctx->d_scene, &ctx->scene, sizeof(SceneArgs), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&ctx->d_frame_ring, (size_t)crt_ctx::EV_RING * sizeof(FrameArgs)));
    CK(hipHostMalloc((void **)&ctx->h_frame_ring, (size_t)crt_ctx::EV_RING * sizeof(FrameArgs)));
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
    if (ctx->d_sync) (void)hipFree(ctx->d_sync);
    for (int i = 0; i < 2; i++) if (ctx->d_rayq[i]) (void)hipFree(ctx->d_rayq[i]);
    if (ctx->d_shadowq) (void)hipFree(ctx->d_shadowq);
    if (ctx->d_occluded) (void)hipFree(ctx->d_occluded);
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_scounts) (void)hipFree(ctx->d_scounts);
    if (ctx->d_exec) (void)hipFree(ctx->d_exec);
    if (ctx->d_heavy) (void)hipFree(ctx->d_heavy);
    if (ctx->d_sheavy) (void)hipFree(ctx->d_sheavy);
    if (ctx->d_hits) (void)hipFree(ctx->d_hits);
    if (ctx->d_hits_all) (void)hipFree(ctx->d_hits_all);
    if (ctx->d_bvh_spill) (void)hipFree(ctx->d_bvh_spill);
    if (ctx->d_lq) (void)hipFree(ctx->d_lq);
    if (ctx->d_lq_words) (void)hipFree(ctx->d_lq_words);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->h_frame_ring) (void)hipHostFree(ctx->h_frame_ring);
    if (ctx->d_frame_ring) (void)hipFree(ctx->d_frame_ring);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    if (ctx->d_fallback_total) (void)hipFree(ctx->d_fallback_total);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    for (int i = 0; i < crt_ctx::EV_RING; i++) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
        if (ctx->ev2[i]) (void)hipEventDestroy(ctx->ev2[i]);
        if (ctx->ev3[i]) (void)hipEventDestroy(ctx->ev3[i]);
        if (ctx->ev4[i]) (void)hipEventDestroy(ctx->ev4[i]);
        if (ctx->ev_fork[i]) (void)hipEventDestroy(ctx->ev_fork[i]);
        if (ctx->ev_s0[i]) (void)hipEventDestroy(ctx->ev_s0[i]);
        if (ctx->ev_s1[i]) (void)hipEventDestroy(ctx->ev_s1[i]);
        if (ctx->ev_s2[i]) (void)hipEventDestroy(ctx->ev_s2[i]);
    }
    if (ctx->ev_call0) (void)hipEventDestroy(ctx->ev_call0);
    if (ctx->ev_call1) (void)hipEventDestroy(ctx->ev_call1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    delete ctx;
}

extern "C" const char *crt_last_error(const crt_ctx *ctx) {
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

extern "C" int crt_set_camera(crt_ctx *ctx, const float position[3], const float matrix[9]) {
    if (!ctx || !position || !matrix) return CRT_ERR_INVALID;
    memcpy(ctx->frame.cam_pos, position, 3 * sizeof(float));
    memcpy(ctx->frame.cam, matrix, 9 * sizeof(float));
    return CRT_OK;
}

static int check_options(crt_ctx *ctx, const crt_options *o) {
    if (!o) { ctx->error = "options is NULL"; return CRT_ERR_INVALID; }
    if (o->use_gi && o->collect_counters == 2) { ctx->error = "collect_counters == 2 belongs to the ray-stream kernels; the GI mode renders with render_lanes"; return CRT_ERR_INVALID; }
    if (o->use_gi && (o->gi_sample_size > 64u || o->rays_per_pixel > 65536u)) { ctx->error = "gi_sample_size > 64 or rays_per_pixel > 65536"; return CRT_ERR_INVALID; }
    if (o->max_depth > 4096) { ctx->error = "max_depth too large"; return CRT_ERR_INVALID; }
    return CRT_OK;
}

static int ensure_items(crt_ctx *ctx, size_t n) {
    if (n <= ctx->items_cap) return CRT_OK;
    CRT_HIP_CHECK(ctx, hipDeviceSynchronize());  // nothing may still be reading the old items
    if (ctx->d_items) (void)hipFree(ctx->d_items);
    ctx->d_items = nullptr;
    ctx->items_cap = 0;
    CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_items, (n + 4) * sizeof(WorkItem)));  // (+: padding, so that a 16-byte read at the last record stays inside)
    ctx->items_cap = n;
    ctx->cached_rects.clear();
    ctx->cached_is_partition = false;
    return CRT_OK;
}

static int ensure_frames(crt_ctx *ctx, uint32_t max_depth, bool gi) {
    const size_t waves = (size_t)ctx->grid_blocks * (BLOCK / 64);
    const size_t per_wave = (size_t)(max_depth + 1) * (gi ? FRAME_DWORDS_GI : FRAME_DWORDS) * 64;
    if (waves * per_wave > ctx->frames_floats) {
        if (ctx->d_frames) (void)hipFree(ctx->d_frames);
        ctx->d_frames = nullptr;
        ctx->frames_floats = 0;
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_frames, waves * per_wave * sizeof(float)));
        ctx->frames_floats = waves * per_wave;
    }
    ctx->frame.frames = ctx->d_frames;
    ctx->frame.frame_wave_stride = per_wave;
    return CRT_OK;
}

// What the finished frames left behind.  Every frame copies its counter block and the fallback total into its own slot of
// the pinned ring (launch_render); a slot is read here only once the frame's last event has completed, so the host never
// reads a buffer a copy may still be writing, and what the next launch learns is a function of a COMPLETED frame.
static void harvest_counts(crt_ctx *ctx) {
    while (ctx->next_count_harvest < ctx->launches) {
        const uint64_t j = ctx->next_count_harvest;
        if (j + crt_ctx::EV_RING <= ctx->launches) { ctx->next_count_harvest++; continue; }  // its slot has been reused
        const int slot = (int)(j % crt_ctx::EV_RING);
        if (hipEventQuery(ctx->ev4[slot]) != hipSuccess) { (void)hipGetLastError(); break; }
        const uint32_t *h = ctx->h_ring + (size_t)slot * crt_ctx::H_SLOT_WORDS;
        if (ctx->slot_items[slot]) {  // (a frame without work items launched nothing and copied nothing)
            ctx->last_counts.assign(h, h + SC_ALLOC_WORDS);
            ctx->last_counts_items = ctx->slot_items[slot];
            ctx->last_counts_cfg = ctx->slot_cfg[slot];
            ctx->fallbacks_seen = h[SC_ALLOC_WORDS];
        }
        ctx->next_count_harvest++;
    }
}

// Queues of the ray-stream pass.  What a frame needs depends on the scene and the camera -- a frame of diffuse surfaces
// has no child rays at all, nested glass can reach 2^(MAX_DEPTH+1)-1 rays per pixel -- so the capacities FOLLOW the frames:
// they start at px * {4 ray-tree nodes, 2 rays per level, n_lights * 2 shadow rays} (px = 64 pixels per work item),
// grow by half when the last completed frame used more than 70 % of one of them, and are regrown inside the call when an attempt
// overflows them (launch_render).  A frame that overflows queues an earlier frame had fitted is not lost either: its queues raise
// the overflow word and render_lanes, which needs no queues, redoes it in the same call (crt_stats::fallback_frames counts
// those; the factors double for the next frame).
static void adapt_queue_sizing(crt_ctx *ctx) {
    const uint32_t *c = ctx->last_counts.data();
    if (ctx->fallbacks_seen != ctx->sizing_seen_fallbacks) {
        ctx->sizing_seen_fallbacks = ctx->fallbacks_seen;
        ctx->node_mult = std::min(4096.0, ctx->node_mult * 2.0);
        ctx->ray_mult = std::min(4096.0, ctx->ray_mult * 2.0);
        ctx->shadow_extra = std::min(4096.0, ctx->shadow_extra * 2.0);
        return;
    }
    const FrameArgs &A = ctx->frame;
    if (!A.s_node_cap || !ctx->last_counts_items) return;
    const uint64_t px = (uint64_t)ctx->last_counts_items * 64;
    uint64_t rays = 0, nodes = px;
    for (int g = 1; g < MAX_GENERATIONS; g++) { rays = std::max<uint64_t>(rays, c[SC_COUNT + g]); nodes += c[SC_COUNT + g]; }
    const uint64_t shadow = c[SC_SHADOW];
    // level 0 owns one node per pixel and n_lights fixed shadow slots per pixel: what can run out is the part beyond that
    const uint64_t base_shadow = px * (ctx->n_lights ? ctx->n_lights : 1);
    const uint64_t extra_nodes = nodes > px ? nodes - px : 0, extra_shadow = shadow > base_shadow ? shadow - base_shadow : 0;
    if (A.s_node_cap > px && extra_nodes * 10 > ((uint64_t)A.s_node_cap - px) * 7) ctx->node_mult = std::min(4096.0, ctx->node_mult * 1.5);
    if (rays * 10 > (uint64_t)A.s_ray_cap * 7) ctx->ray_mult = std::min(4096.0, ctx->ray_mult * 1.5);
    if (A.s_shadow_cap > base_shadow && extra_shadow * 10 > ((uint64_t)A.s_shadow_cap - base_shadow) * 7)
        ctx->shadow_extra = std::min(4096.0, ctx->shadow_extra * 1.5);
}

static uint64_t queue_bytes_for(const crt_ctx *ctx, uint64_t px, double node_mult, double ray_mult, double shadow_extra) {
    const uint64_t lights = ctx->n_lights ? ctx->n_lights : 1;
    return (uint64_t)(px * std::max(ray_mult, 1.0)) * 16 + (uint64_t)(px * ray_mult) * (64 + 24) + (uint64_t)(px * lights * (1.0 + shadow_extra)) * 33 + (uint64_t)(px * node_mult) * 32;
}

// After an attempt that overflowed (launch_render): capacities from what the attempt learnt.  The level that was emitting when a
// queue ran out still counted every ray it wanted to queue (the counters are bumped before the capacity check), so the rays of
// levels 0 .. G are known exactly; the deeper ones are extrapolated with the last growth ratio (a GI frame grows by up to
// gi_samples + 1 per level, a mirror room by 2, most frames shrink), the shadow rays in proportion to the nodes.  25 % on top;
// false when the device has no room for that.
static bool grow_queue_sizing(crt_ctx *ctx, uint32_t vitems, uint32_t max_depth) {
    const uint32_t *c = ctx->last_counts.data();
    const double px = (double)vitems * 64.0, lights = ctx->n_lights ? ctx->n_lights : 1;
    uint32_t G = 0;
    for (uint32_t g = 1; g <= max_depth && g < (uint32_t)MAX_GENERATIONS; g++) if (c[SC_COUNT + g]) G = g;
    double known = px, widest = 0, last = px, before = px;
    for (uint32_t g = 1; g <= G; g++) { before = last; last = c[SC_COUNT + g]; known += last; widest = std::max(widest, last); }
    const double ratio = G >= 1 ? std::max(1.0, last / std::max(before, 1.0)) : 2.0;
    double total = known, level = last;
    for (uint32_t g = G + 1; g <= max_depth; g++) { level *= ratio; total += level; widest = std::max(widest, level); }
    const double shadows = (double)c[SC_SHADOW] * (total / known);
    double node_mult = std::max(ctx->node_mult, total * 1.25 / px), ray_mult = std::max(ctx->ray_mult, widest * 1.25 / px);
    double shadow_extra = std::max(ctx->shadow_extra, shadows * 1.25 / (px * lights) - 1.0);
    if (node_mult == ctx->node_mult && ray_mult == ctx->ray_mult && shadow_extra == ctx->shadow_extra) {
        node_mult *= 2.0; ray_mult *= 2.0; shadow_extra = std::max(0.5, shadow_extra * 2.0);  // (the counters explain nothing: an eviction list, say)
    }
    if (px * node_mult > 2.0e9 || px * ray_mult > 2.0e9 || px * lights * (1.0 + shadow_extra) > 2.0e9) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (queue_bytes_for(ctx, (uint64_t)px, node_mult, ray_mult, shadow_extra) > (free_b + ctx->queue_bytes) / 2) return false;
    ctx->node_mult = node_mult; ctx->ray_mult = ray_mult; ctx->shadow_extra = shadow_extra;
    return true;
}

static int ensure_stream(crt_ctx *ctx, uint32_t n_items) {
    FrameArgs &A = ctx->frame;
    adapt_queue_sizing(ctx);
    const uint64_t px = (uint64_t)n_items * 64;
    const uint64_t lights = ctx->n_lights ? ctx->n_lights : 1;
    const uint64_t floor_cap = 1u << 16;
    uint64_t node_cap = std::max<uint64_t>(floor_cap, (uint64_t)(px * ctx->node_mult));
    uint64_t ray_cap = std::max<uint64_t>(floor_cap, (uint64_t)(px * ctx->ray_mult));
    uint64_t shadow_cap = std::max<uint64_t>(floor_cap, (uint64_t)(px * lights * (1.0 + ctx->shadow_extra)));
    // explicit capacities (crt_tuning): never below what level 0 itself needs, so that only the deeper levels can overflow
    if (ctx->tuning.node_cap) node_cap = ctx->tuning.node_cap < px ? px : ctx->tuning.node_cap;
    if (ctx->tuning.ray_cap) ray_cap = ctx->tuning.ray_cap;
    if (ctx->tuning.shadow_cap) shadow_cap = ctx->tuning.shadow_cap;
    node_cap = std::min<uint64_t>(node_cap, 0x7FFFFFF0ull);
    ray_cap = std::min<uint64_t>(ray_cap, 0x7FFFFFF0ull);
    shadow_cap = std::min<uint64_t>(shadow_cap, 0x7FFFFFF0ull);
    if (px > node_cap) { ctx->error = "frame too large for the ray-stream buffers"; return CRT_ERR_INVALID; }
    const bool fixed_caps = ctx->tuning.node_cap || ctx->tuning.ray_cap || ctx->tuning.shadow_cap;
    const bool grow = node_cap > A.s_node_cap || ray_cap > A.s_ray_cap || shadow_cap > A.s_shadow_cap || n_items > ctx->stream_items ||
                      (fixed_caps && (node_cap != A.s_node_cap || ray_cap != A.s_ray_cap || shadow_cap != A.s_shadow_cap));
    if (grow) {
        CRT_HIP_CHECK(ctx, hipDeviceSynchronize());  // nothing may still be using the old buffers
        if (!fixed_caps) {  // never shrink: keep what is already there
            node_cap = std::max<uint64_t>(node_cap, A.s_node_cap);
            ray_cap = std::max<uint64_t>(ray_cap, A.s_ray_cap);
            shadow_cap = std::max<uint64_t>(shadow_cap, A.s_shadow_cap);
        }
        void **bufs[] = {(void **)&ctx->d_rayq[0], (void **)&ctx->d_rayq[1], (void **)&ctx->d_shadowq, (void **)&ctx->d_occluded,
                         (void **)&ctx->d_nodes, (void **)&ctx->d_heavy, (void **)&ctx->d_sheavy, (void **)&ctx->d_hits, (void **)&ctx->d_hits_all,
                         (void **)&ctx->d_lq};
        for (void **b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
        ctx->stream_items = 0;
        A.s_node_cap = A.s_ray_cap = A.s_shadow_cap = 0;
        // (+ 64 bytes of padding behind the queues)
        for (int i = 0; i < 2; i++) CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_rayq[i], ray_cap * 2 * sizeof(float4) + 64));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_shadowq, shadow_cap * 2 * sizeof(float4) + 64));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_occluded, shadow_cap));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_nodes, node_cap * 2 * sizeof(float4)));
        ctx->heavy_cap = (uint32_t)std::max<uint64_t>(floor_cap, ray_cap);  // (a full list only keeps a long walk where it is)
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_heavy, (size_t)ctx->heavy_cap * sizeof(uint32_t) + 64));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_sheavy, (size_t)ctx->heavy_cap * sizeof(uint32_t) + 64));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_hits, (size_t)ctx->heavy_cap * sizeof(float4)));
        CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_hits_all, (size_t)std::max<uint64_t>(ray_cap, px) * sizeof(float4)));  // (level 0 holds px rays, a deeper level at most ray_cap)
        if (ctx->scene.bvh_ok && ctx->tuning.bvh && ctx->tuning.level_queue) {
            // the level queue: every ray below level 0 owns a node, so node_cap entries always do; its tags start as "no frame's"
            CRT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_lq, node_cap * 8 * sizeof(unsigned long long)));
            CRT_HIP_CHECK(ctx, hipMemset(ctx->d_lq, 0, node_cap * 8 * sizeof(unsigned long long)));
        }
        A.s_ray_cap = (uint32_t)ray_cap; A.s_shadow_cap = (uint32_t)shadow_cap; A.s_node_cap = (uint32_t)node_cap;
        ctx->stream_items = n_items;
        ctx->queue_bytes = ray_cap * 64 + shadow_cap * 33 + node_cap * 32 + (size_t)ctx->heavy_cap * 24 + std::max<uint64_t>(ray_cap, px) * 16 + (ctx->d_lq ? node_cap * 64 : 0);
    }
    A.s_rayq[0] = ctx->d_rayq[0]; A.s_rayq[1] = ctx->d_rayq[1];
    A.s_shadowq = ctx->d_shadowq; A.s_occluded = ctx->d_occluded; A.s_nodes = ctx->d_nodes;
    A.s_heavy = ctx->d_heavy; A.s_sheavy = ctx->d_sheavy; A.s_hits = ctx->d_hits; A.s_hits_all = ctx->d_hits_all; A.s_heavy_cap = ctx->heavy_cap;
    // (an explicit ray capacity -- crt_tuning, the tests' way to an overflow -- bounds the level queue too)
    A.s_lq = ctx->d_lq; A.s_lq_cap = ctx->d_lq ? (ctx->tuning.ray_cap ? std::min(A.s_ray_cap, A.s_node_cap) : A.s_node_cap) : 0u;
    return CRT_OK;
}

// level 0 of the ray-stream pass holds this many rays per pixel (RayTracer.cpp:90-104: the centre sample + RAYS_PER_PIXEL - 1 jittered ones)
static uint32_t level0_samples_of(const crt_options *o) { return o->use_gi && o->rays_per_pixel > 1u ? o->rays_per_pixel : 1u; }
// a GI frame goes through the ray-stream pass when its level 0 fits 31-bit ray indices (otherwise render_lanes<.., true> renders it pixel by pixel)
static bool gi_fits_stream(const crt_options *o, uint32_t n_items) {
    return (uint64_t)n_items * 64u * level0_samples_of(o) * 2u < (1ull << 31) && o->gi_sample_size <= 64u;
}
// what, beside its size, decides how many rays a frame queues: two frames are "of the same kind" for the queue sizing when this agrees
static uint32_t frame_config_of(const crt_options *o) {
    return (o->use_gi ? 0x80000000u : 0u) | ((o->use_gi ? o->gi_sample_size & 0x7Fu : 0u) << 24) | (o->max_depth & 0xFFFFFFu);
}

static int g_debug_sync = 0;   // development (crt_tuning::bvh == 3): announce every launch of launch() on stderr and wait for it
template <typename K, typename... Args>
static void launch(K kernel, uint32_t blocks, hipStream_t stream, Args... args) {
    static int serial = 0;
    if (g_debug_sync) { fprintf(stderr, "[launch %d] kernel %p blocks %u ...", serial, (void *)kernel, blocks); fflush(stderr); }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK), 0, stream, args...);
    if (g_debug_sync) { hipError_t e = hipDeviceSynchronize(); fprintf(stderr, " done (%s)\n", hipGetErrorString(e)); fflush(stderr); serial++; }
}
template <typename K, typename... Args>
static void launch_lds(K kernel, uint32_t blocks, uint32_t lds_bytes, hipStream_t stream, Args... args) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK), lds_bytes, stream, args...);
}

// What launch_frame decides once per frame and its three launch helpers read.
struct FramePlan {
    int slot;                     // event / pinned-counter / argument-block slot of this frame
    uint32_t n_items, vitems;     // work items; work items' worth of level-0 rays (the GI mode: x rays_per_pixel)
    uint32_t lane_blocks;         // grid of the kernels that take one ray (or pixel) per lane
    bool gi, count, exec_count;   // GI mode; counting build; production kernels tallying the tests they execute
    bool heavy, lean, wide;       // wave-per-ray kernels on; plan kernels; the wide plan
    bool bvh;                     // the filter kernels (kernel_bvh.h) walk the rays
    bool queue;                   // ... and every level below level 0 is one launch (bvh_trace_queue)
    uint32_t level_budget;        // steps after which a deeper level's per-lane walk is evicted
    const uint32_t *prev;         // counters of a completed frame of this size and kind, or null
    bool last_resort;             // render_lanes behind the stream pass
};

// Levels 0 .. MAX_DEPTH on `stream`; after level 0 the bulk shadow pass (and the walks it gives up) on the side stream.
static int launch_stream_levels(crt_ctx *ctx, const crt_options *o, FramePlan &P, KernelArgs &A, hipStream_t stream) {
    CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_scounts, 0, SC_ALLOC_WORDS * sizeof(uint32_t), stream));
    if (P.queue) for (int w : {LQ_TAIL, LQ_HEAD, LQ_DONE, LQ_ABORT}) CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_lq_words + w, 0, 16, stream));
    const uint32_t side_per_cu = ctx->tuning.side_blocks;  // workgroups per CU of the bulk shadow pass beside the levels
    A.exec_count = P.exec_count ? 1u : 0u;
    A.exec_counters = ctx->d_exec;
    A.exec_plan = ctx->d_exec + 4;
    if (P.exec_count) CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_exec, 0, 6 * sizeof(unsigned long long), stream));
    // the plan kernels pay a wave-uniform loop per refill, whatever the number of new rays: refill in bundles
    A.bundle = REFILL_BUNDLE;
    A.wave_prio = 3u;  // the levels' waves (the frame's critical path) ahead of the bulk shadow pass's, which share their SIMDs
    A.force_whole = 0u;
    if (ctx->frame.fixed0) CRT_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->d_scounts + SC_SHADOW), (int)((uint32_t)P.vitems * 64u * ctx->n_lights), 1, stream));
    const uint32_t plds = ctx->scene.plan_list_words * BLOCK * (uint32_t)sizeof(uint32_t);  // kernel_plan.h: mesh lists
    KernelArgs S = A;  // argument block of the bulk shadow pass
    S.wave_prio = 0u;
    S.counters = ctx->d_counters + C_N;
    S.exec_counters = ctx->d_exec + 2;  // it tallies on its own
    S.exec_plan = ctx->d_exec + 5;
    {
        // The pass is one persistent launch: it ends when its longest walk ends, so the budget after which a walk is
        // handed to heavy_trace_shadow should be about the steps one lane gets through in the whole launch --
        // rays per lane x ~130 steps per ray (measured average on the benchmark scenes) -- and no more than the cap
        // (crt_tuning::shadow_budget).  A rank that renders 1/8 of the tiles gets 1/8 of the budget.
        const uint64_t lanes = (uint64_t)ctx->num_cus * (side_per_cu ? side_per_cu : 8u) * BLOCK;
        const uint64_t est = (uint64_t)P.vitems * 64u * (ctx->n_lights ? ctx->n_lights : 1u) * 130u / (lanes ? lanes : 1u);
        uint32_t budget = est > ctx->tuning.shadow_budget ? ctx->tuning.shadow_budget : (uint32_t)est;
        if (budget < ctx->step_budget) budget = std::min(ctx->step_budget, ctx->tuning.shadow_budget);
        S.step_budget = P.heavy ? budget : 0u;
    }
    // a completed frame of this size (one that overflowed stopped early: its levels' counts say nothing)
    const uint32_t *prev = ctx->last_counts_items == P.vitems && ctx->last_counts_cfg == frame_config_of(o) && !ctx->last_counts[SC_OVERFLOW] ? ctx->last_counts.data() : nullptr;
    // (under the P.wide plan a ray crosses dozens of small mesh trees: a walk of a thousand steps is the rule there, not the outlier the
    //  wave-per-ray kernel is for -- measured on tools/many_meshes.py 200: 32.7 ms per frame with the plain budget, 25.4 with four times it)
    // (the same holds for a GI frame's levels -- millions of incoherent rays each: tools/gi_time.py hw14 960x540 d3 n2 r2 54.7 ms -> 44.2)
    const uint32_t level_budget = (P.wide || P.gi) ? std::min<uint32_t>(ctx->step_budget * 4u, 1u << 20) : ctx->step_budget;
    P.prev = prev; P.level_budget = level_budget;  // (launch_stream_tail sizes its pass by them too)
    // the same reasoning for level 0 (one launch over all primary rays, ~70 steps per ray)
    const uint64_t est0 = (uint64_t)P.vitems * 64u * 70u / ((uint64_t)P.lane_blocks * BLOCK);
    uint32_t budget0 = est0 >= level_budget ? level_budget : (est0 < 64u ? 64u : (uint32_t)est0);
    if (ctx->tuning.level0_budget) budget0 = ctx->tuning.level0_budget;
    for (uint32_t g = 0; g <= o->max_depth; g++) {
        A.step_budget = P.heavy ? (g == 0 ? budget0 : level_budget) : 0u;
        // The per-lane kernel of a deeper level fetches its rays through a cursor, so any grid does the whole level; beside
        // the bulk shadow pass every workgroup of it waits for a free P.slot, and a level below heavy_level_threshold has
        // nothing for it to do (measured: 0.15 ms for an empty full-size grid).  Sized by what the level held a frame ago.
        uint32_t level_blocks = P.lane_blocks;
        if (g >= 1 && P.lean && prev) {
            const uint32_t was = prev[SC_COUNT + g];
            const uint32_t want = was < ctx->frame.heavy_level_threshold ? 64u : std::max<uint32_t>((uint32_t)ctx->num_cus, (was + was / 2u + BLOCK - 1) / BLOCK);
            level_blocks = std::min(P.lane_blocks, want);
        }
        if (P.queue && g == 0) {
            if (P.exec_count) launch(bvh_trace_level0<BVH_TALLY>, P.lane_blocks, stream, A);
            else if (ctx->tuning.bvh == 2) launch(bvh_trace_level0<BVH_CHECKED>, P.lane_blocks, stream, A);
            else launch(bvh_trace_level0<BVH_PLAIN>, P.lane_blocks, stream, A);
        }
        else if (P.bvh) launch(P.exec_count ? bvh_trace_shade_tally : ctx->tuning.bvh == 2 ? bvh_trace_shade_checked : bvh_trace_shade, g == 0 ? P.lane_blocks : level_blocks, stream, A, g);
        else if (P.count) { if (P.gi) launch(stream_trace_shade<true, true>, P.lane_blocks, stream, A, g); else launch(stream_trace_shade<true>, P.lane_blocks, stream, A, g); }
        else if (P.lean && P.gi) {
            // the GI mode: the walk alone, then the level's shading -- sample directions, gi_samples child rays -- with every lane busy
            // (kernel_plan.h, SPLIT; tools/gi_time.py hw14 960x540 d3 n2 r2: 42.3 ms on the device against 44.0 with the shading inside the walk)
            launch_lds(P.wide ? stream_trace_plan_wide_gi : stream_trace_plan_gi, level_blocks, plds, stream, A, g);
            launch(stream_shade_all<true>, level_blocks, stream, A, g);
        }
        else if (P.wide) launch_lds(stream_trace_shade_plan_wide, level_blocks, plds, stream, A, g);
        else if (P.lean) launch_lds(stream_trace_shade_plan, level_blocks, plds, stream, A, g);
        else if (P.gi) launch(stream_trace_shade<false, true>, P.lane_blocks, stream, A, g);
        else launch(stream_trace_shade<false>, P.lane_blocks, stream, A, g);
        if (P.heavy && !P.bvh) {   // (the filter kernels walk what they cannot decide themselves, in the reference's order: nothing is handed over)
            launch(P.gi ? heavy_trace_closest_gi : P.exec_count ? heavy_trace_closest_tally : heavy_trace_closest, HEAVY_BLOCKS, stream, A, g);
            if (P.gi) launch(stream_shade_evicted<false, true>, 256u, stream, A, g);
            else launch(stream_shade_evicted<false>, 256u, stream, A, g);
        }
        if (g == 0) {
            // where level 0's shadow rays end; they start now, on the side stream, beside the deeper levels
            hipLaunchKernelGGL(stream_mark_split, dim3(1), dim3(64), 0, stream, A, (uint32_t)SC_SHADOW_SPLIT, (uint32_t)SC_SHADOW);
            hipStream_t where = side_per_cu ? ctx->side : stream;
            if (side_per_cu) {
                CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork[P.slot], stream));
                CRT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork[P.slot], 0));
            }
            CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_s0[P.slot], where));
            // (beside the levels its persistent waves must leave wave slots on every CU for the level kernels)
            const uint32_t blocks0 = side_per_cu ? (uint32_t)ctx->num_cus * side_per_cu : ctx->grid_blocks;
            if (P.bvh) {
                if (P.exec_count) launch(bvh_trace_shadow<0, BVH_TALLY>, blocks0, where, S);
                else if (ctx->tuning.bvh == 2) launch(bvh_trace_shadow<0, BVH_CHECKED>, blocks0, where, S);
                else launch(bvh_trace_shadow<0, BVH_PLAIN>, blocks0, where, S);
            }
            else if (P.count) launch(stream_trace_shadow<true>, blocks0, where, S, 0u);
            else if (P.wide) launch(stream_trace_shadow_plan_wide<0>, blocks0, where, S);
            else if (P.lean) launch(stream_trace_shadow_plan<0>, blocks0, where, S);
            else launch(stream_trace_shadow<false>, blocks0, where, S, 0u);
            // ... and behind it the walks it gave up, still beside the levels; the mark comes before the event the
            // caller's stream waits for, so nothing the later pass appends is below it
            if (P.heavy && !P.bvh) hipLaunchKernelGGL(stream_mark_split, dim3(1), dim3(64), 0, where, S, (uint32_t)SC_SHEAVY_SPLIT, (uint32_t)SC_SHEAVY);
            CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_s1[P.slot], where));
            if (P.heavy && !P.bvh) launch(P.exec_count ? heavy_trace_shadow_tally : heavy_trace_shadow, HEAVY_BLOCKS, where, S, 0u);
            CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_s2[P.slot], where));
        }
        if (P.queue) {
            // every level below level 0: one launch that feeds itself (kernel_bvh.h); two workgroups per CU hold more lanes than the
            // widest level of a frame of this size has rays
            const uint32_t qblocks = std::min(P.lane_blocks, (uint32_t)ctx->num_cus * std::max(1u, ctx->tuning.level_queue));
            if (o->max_depth >= 1) {
                A.bundle = ctx->tuning.level0_budget ? ctx->tuning.level0_budget - 1u : 15u;   // (development: turns between two housekeeping rounds, as a mask)
                if (P.exec_count) launch(bvh_trace_queue<BVH_TALLY>, qblocks, stream, A);
                else if (ctx->tuning.bvh == 2) launch(bvh_trace_queue<BVH_CHECKED>, qblocks, stream, A);
                else launch(bvh_trace_queue<BVH_PLAIN>, qblocks, stream, A);
            }
            break;
        }
    }
    return CRT_OK;
}

// The deeper levels' shadow rays, the wave-per-ray walks of both passes, the per-pixel combination, the last resort.
static int launch_stream_tail(crt_ctx *ctx, FramePlan &P, KernelArgs &A, hipStream_t stream) {
    const uint32_t side_per_cu = ctx->tuning.side_blocks;
    const uint32_t *prev = P.prev;
    const uint32_t level_budget = P.level_budget;
    CRT_HIP_CHECK(ctx, hipGetLastError());
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1[P.slot], stream));
    // the shadow rays of the deeper levels (queued behind level 0's), then the wave-per-ray walks of both passes
    if (side_per_cu) CRT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_s1[P.slot], 0));
    KernelArgs S1 = A;
    S1.wave_prio = 0u;
    S1.counters = ctx->d_counters + 2 * C_N;
    {
        // few rays, all tail: the short budget of the levels, or less when this launch is small (the deeper levels
        // queue about a quarter of a shadow ray per pixel on the benchmark scenes)
        const uint64_t rays1 = prev && prev[SC_SHADOW] >= prev[SC_SHADOW_SPLIT] ? prev[SC_SHADOW] - prev[SC_SHADOW_SPLIT] : (uint64_t)P.vitems * 16u;
        const uint64_t est1 = rays1 * 130u / ((uint64_t)P.lane_blocks * BLOCK);
        // (a GI frame queues most of its shadow rays here -- tens of millions: then this pass is a bulk pass like pass 0 and gets its cap)
        const uint32_t cap1 = std::max(level_budget, ctx->tuning.shadow_budget);
        S1.step_budget = P.heavy ? (est1 >= cap1 ? cap1 : (est1 < 64u ? 64u : (uint32_t)est1)) : 0u;
    }
    if (P.bvh) {
        if (P.exec_count) launch(bvh_trace_shadow<1, BVH_TALLY>, P.lane_blocks, stream, S1);
        else if (ctx->tuning.bvh == 2) launch(bvh_trace_shadow<1, BVH_CHECKED>, P.lane_blocks, stream, S1);
        else launch(bvh_trace_shadow<1, BVH_PLAIN>, P.lane_blocks, stream, S1);
    }
    else if (P.count) launch(stream_trace_shadow<true>, P.lane_blocks, stream, S1, 1u);
    else if (P.wide) launch(stream_trace_shadow_plan_wide<1>, P.lane_blocks, stream, S1);
    else if (P.lean) launch(stream_trace_shadow_plan<1>, P.lane_blocks, stream, S1);
    else launch(stream_trace_shadow<false>, P.lane_blocks, stream, S1, 1u);
    if (side_per_cu) CRT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_s2[P.slot], 0));
    if (P.heavy && !P.bvh) launch(P.exec_count ? heavy_trace_shadow_tally : heavy_trace_shadow, HEAVY_BLOCKS, stream, S1, 1u);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev2[P.slot], stream));
    A.counters = ctx->d_counters + 2 * C_N;
    // post-order combination per pixel, then the queue-less fallback, which only runs after an overflow
    if (P.count) launch(stream_resolve<true>, P.lane_blocks, stream, A);
    else launch(stream_resolve<false>, P.lane_blocks, stream, A);
    A.only_if_overflow = 1;
    const uint32_t fallback_blocks = std::max(1u, std::min(ctx->grid_blocks, (P.n_items * 64u + BLOCK - 1) / BLOCK));
    if (!P.last_resort) {}  // a probing attempt (launch_render): the host looks at the overflow word itself
    else if (P.gi && P.count) launch(render_lanes<true, true>, fallback_blocks, stream, A);
    else if (P.gi) launch(render_lanes<false, true>, fallback_blocks, stream, A);
    else if (P.count) launch(render_lanes<true>, fallback_blocks, stream, A);
    else launch(render_lanes<false>, fallback_blocks, stream, A);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    return CRT_OK;
}

// The whole frame by render_lanes: crt_tuning::mode = lanes, and GI frames too large for the ray-stream pass.
static int launch_lanes_pass(crt_ctx *ctx, FramePlan &P, KernelArgs &A, hipStream_t stream) {
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1[P.slot], stream));
    A.counters = ctx->d_counters + C_N;
    if (P.gi && P.count) launch(render_lanes<true, true>, P.lane_blocks, stream, A);
    else if (P.gi) launch(render_lanes<false, true>, P.lane_blocks, stream, A);
    else if (P.count) launch(render_lanes<true>, P.lane_blocks, stream, A);
    else launch(render_lanes<false>, P.lane_blocks, stream, A);
    CRT_HIP_CHECK(ctx, hipGetLastError());
    for (hipEvent_t e : {ctx->ev_s0[P.slot], ctx->ev_s1[P.slot], ctx->ev_s2[P.slot], ctx->ev2[P.slot]}) CRT_HIP_CHECK(ctx, hipEventRecord(e, stream));
    return CRT_OK;
}

// One frame's launches.  Ray-stream path (kernel_stream.h), per recursion level g = 0 .. MAX_DEPTH on `stream`:
//   the per-lane kernel (plan kernels; the faithful kernel for the counting build and for scenes without a plan),
//   heavy_trace_closest for the walks it handed over (or the whole level), stream_shade_evicted for their hits;
// the bulk shadow pass (level 0's shadow rays) on the side stream as soon as level 0 is done, then the deeper levels'
// shadow rays, the wave-per-ray shadow walks, stream_resolve, and render_lanes, which only runs after a queue overflow.
static int launch_frame(crt_ctx *ctx, const crt_options *o, uint32_t n_items, float *d_out, uint32_t packed, hipStream_t stream, bool last_resort) {
    const bool gi = o->use_gi != 0;  // the GI / multi-sample mode: rendered pixel by pixel by render_lanes<.., true> (kernel_lane.h)
    int rc = ensure_frames(ctx, o->max_depth, gi);
    if (rc) return rc;
    const SceneArgs &SC = ctx->scene;
    FrameArgs &F = ctx->frame;
    F.use_gi = gi ? 1u : 0u;
    F.gi_samples = o->gi_sample_size;
    F.rays_per_pixel = o->rays_per_pixel;
    F.monte_carlo_bias = o->monte_carlo_bias;
    F.gi_seed = o->gi_seed;
    // the GI mode's level 0 holds rays_per_pixel rays per pixel: `vitems` work items' worth of rays, which is what queues and grids are sized by
    const uint32_t samples = level0_samples_of(o);
    const bool stream_mode = ctx->mode == crt_ctx::MODE_STREAM && (!gi || gi_fits_stream(o, n_items));
    const uint32_t vitems = stream_mode ? n_items * samples : n_items;
    F.level0_samples = stream_mode ? samples : 1u;
    if (stream_mode && o->max_depth + 1 > (uint32_t)MAX_GENERATIONS) {
        ctx->error = "max_depth too large for the ray-stream pass";
        return CRT_ERR_INVALID;
    }
    F.max_depth = o->max_depth;
    F.shadow_bias = o->shadow_bias;
    F.reflection_bias = o->reflection_bias;
    F.refraction_bias = o->refraction_bias;
    F.items = ctx->d_items;
    F.n_items = n_items;
    F.pixel_counter = ctx->d_sync + 0;
    F.out = d_out;
    F.packed = packed;
    F.s_counts = ctx->d_scounts;
    F.fallback_total = ctx->d_fallback_total;
    F.s_lq_words = ctx->d_lq_words;
    F.lq_epoch = (uint32_t)(ctx->launches & 0x7FFFFFFFull) + 1u;   // the level queue's tag of this frame (this context's buffer has never seen it: tags only grow)
    F.bvh_spill = ctx->d_bvh_spill;
    F.bvh_spill_side = ctx->d_bvh_spill ? ctx->d_bvh_spill + (size_t)ctx->grid_blocks * BLOCK * (ctx->scene.bvh_stack - BVH_LDS_STACK) : nullptr;
    const bool count = o->collect_counters == 1;       // the counting build: every ray walked the reference's way
    const bool exec_count = o->collect_counters == 2;  // the production kernels, tallying the tests they execute
    CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_sync, 0, 4 * sizeof(uint32_t), stream));
    if (count) CRT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, 3 * C_N * sizeof(unsigned long long), stream));
    harvest_counts(ctx);
    const int slot = (int)(ctx->launches % crt_ctx::EV_RING);
    if (ctx->launches >= (uint64_t)crt_ctx::EV_RING) {
        // the slot's previous frame (64 launches ago) must be complete before its events, its pinned words and its argument block are reused
        CRT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev4[slot]));
        harvest_counts(ctx);
    }
    ctx->slot_items[slot] = 0;
    const uint32_t lane_need = (vitems * 64u + BLOCK - 1) / BLOCK;
    const uint32_t lane_blocks = std::max(1u, lane_need < ctx->grid_blocks ? lane_need : ctx->grid_blocks);
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0[slot], stream));
    if (n_items == 0) {
        // nothing to render: the events still bracket an (empty) frame, so that every reader of a slot finds recorded events
        for (hipEvent_t e : {ctx->ev1[slot], ctx->ev_s0[slot], ctx->ev_s1[slot], ctx->ev_s2[slot], ctx->ev2[slot], ctx->ev3[slot], ctx->ev4[slot]})
            CRT_HIP_CHECK(ctx, hipEventRecord(e, stream));
        ctx->launches++;
        return CRT_OK;
    }
    // the wave-per-ray path needs nested boxes; the counting build walks every ray the reference's way
    const bool heavy = stream_mode && ctx->step_budget && SC.nested_boxes && (SC.top_fast || SC.plan_seq) && !count;
    // the plan kernels (kernel_plan.h): a small top-level tree (its leaves as a plan), 32-bit offsets, compact leaf links
    const bool lean = heavy && ctx->lean_ok && (SC.plan_ok || SC.plan_wide);
    const bool wide = lean && !SC.plan_ok;  // the wide plan (kernel_plan.h): more than 64 top-level leaves or meshes
    // the filter kernels (kernel_bvh.h): whenever the scene has a filter -- they need nothing of the plan or of the wave-per-ray kernels
    const bool bvh = stream_mode && SC.bvh_ok && ctx->tuning.bvh && !gi && !count;
    const bool queue = bvh && ctx->tuning.level_queue;
    g_debug_sync = ctx->tuning.bvh == 3 ? 1 : 0;
    if (g_debug_sync) fprintf(stderr, "[frame] bvh_trace_shade %p tally %p checked %p shadow0 %p shadow1 %p heavy_closest %p shade_evicted %p resolve %p heavy_shadow %p\n",
                              (void *)bvh_trace_shade, (void *)bvh_trace_shade_tally, (void *)bvh_trace_shade_checked, (void *)bvh_trace_shadow<0, BVH_PLAIN>,
                              (void *)bvh_trace_shadow<1, BVH_PLAIN>, (void *)heavy_trace_closest, (void *)stream_shade_evicted<false>, (void *)stream_resolve<false>, (void *)heavy_trace_shadow);
    if (stream_mode) {
        rc = ensure_stream(ctx, vitems);
        if (rc) return rc;
        F.heavy_level_threshold = (lean && !bvh) ? ctx->tuning.heavy_level : 0u;  // (filter frames: the wave-per-ray kernel takes the evicted rays only)
        // level 0 owns the first n_items * 64 * n_lights slots of the shadow queue; the deeper levels append
        F.fixed0 = (uint64_t)vitems * 64u * ctx->n_lights <= F.s_shadow_cap ? 1u : 0u;
    }
    // this frame's argument block, into its own slot (the copy is ordered on `stream` ahead of the kernels that read it)
    ctx->h_frame_ring[slot] = F;
    CRT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_frame_ring + slot, ctx->h_frame_ring + slot, sizeof(FrameArgs), hipMemcpyHostToDevice, stream));
    KernelArgs A{};
    A.s = (scene_args_p)ctx->d_scene;
    A.f = (frame_args_p)(ctx->d_frame_ring + slot);
    A.counters = ctx->d_counters;
    FramePlan P{slot, n_items, vitems, lane_blocks, gi, count, exec_count, heavy, lean, wide, bvh, queue, 0u, nullptr, last_resort};
    rc = stream_mode ? launch_stream_levels(ctx, o, P, A, stream) : launch_lanes_pass(ctx, P, A, stream);
    if (rc == CRT_OK && stream_mode) rc = launch_stream_tail(ctx, P, A, stream);
    if (rc) return rc;
    // what this frame leaves for the next ones: its counter block and the fallback total, into this frame's own pinned slot
    uint32_t *h = ctx->h_ring + (size_t)slot * crt_ctx::H_SLOT_WORDS;
    if (stream_mode) CRT_HIP_CHECK(ctx, hipMemcpyAsync(h, ctx->d_scounts, SC_ALLOC_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CRT_HIP_CHECK(ctx, hipMemcpyAsync(h + SC_ALLOC_WORDS, ctx->d_fallback_total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ctx->slot_items[slot] = stream_mode ? vitems : 0u;
    ctx->slot_cfg[slot] = frame_config_of(o);
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev3[slot], stream));
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev4[slot], stream));
    ctx->launches++;
    return CRT_OK;
}

// A frame whose queues are not yet known to fit -- the first of its size on this context, or the one after a frame that
// overflowed -- is PROBED: the stream pass is enqueued without the queue-less fallback behind it, the call waits for it and
// reads its overflow word, and an attempt that did not fit is repeated with queues sized by what it learnt (grow_queue_sizing) until it
// does; every attempt ends early at the first overflow, so a failed one costs less than a frame.  Once a frame of this size has
// completed without overflow the call is asynchronous again, with render_lanes behind the stream pass as the last resort for a
// frame that outgrows its queues all the same (and for explicit capacities, crt_tuning, which are never regrown).
static int launch_render(crt_ctx *ctx, const crt_options *o, uint32_t n_items, float *d_out, uint32_t packed, hipStream_t stream) {
    const bool stream_mode = ctx->mode == crt_ctx::MODE_STREAM && (!o->use_gi || gi_fits_stream(o, n_items));
    const bool fixed_caps = ctx->tuning.node_cap || ctx->tuning.ray_cap || ctx->tuning.shadow_cap;
    const uint32_t vitems = stream_mode ? n_items * level0_samples_of(o) : n_items;
    harvest_counts(ctx);
    const bool proven = ctx->last_counts_items == vitems && ctx->last_counts_cfg == frame_config_of(o) && ctx->last_counts[SC_OVERFLOW] == 0;
    if (!stream_mode || fixed_caps || proven || n_items == 0) return launch_frame(ctx, o, n_items, d_out, packed, stream, true);
    for (int attempt = 0;; attempt++) {
        const bool last = attempt == 5;
        const int slot = (int)(ctx->launches % crt_ctx::EV_RING);
        const int rc = launch_frame(ctx, o, n_items, d_out, packed, stream, last);
        if (rc || last) return rc;
        CRT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev4[slot]));
        harvest_counts(ctx);
        if (ctx->last_counts_items != vitems) { ctx->error = "internal: a probing frame left no counters"; return CRT_ERR_HIP; }
        if (!ctx->last_counts[SC_OVERFLOW]) return CRT_OK;
        ctx->regrows++;
        const bool can_grow = !ctx->last_counts[SC_GUARD] && grow_queue_sizing(ctx, vitems, o->max_depth);
        if (!can_grow) return launch_frame(ctx, o, n_items, d_out, packed, stream, true);  // (a walk beyond its bound, or no memory to grow into)
    }
}

// Called after a synchronisation: the last frame's slot tells whether the frame was redone by the queue-less kernel.
static void note_overflow(crt_ctx *ctx) {
    harvest_counts(ctx);
    ctx->overflows = ctx->fallbacks_seen;
    ctx->stats.fallback_frames = (uint32_t)ctx->overflows;
    ctx->stats.queue_bytes = ctx->queue_bytes;
    ctx->stats.queue_regrows = ctx->regrows;
}

static int fetch_counters(crt_ctx *ctx, const crt_options *o, uint64_t pixels) {
    ctx->stats.pixels = pixels;
    ctx->stats.counters_valid = o->collect_counters == 1 ? 1 : 0;
    if (o->collect_counters == 2)
        CRT_HIP_CHECK(ctx, hipMemcpy(ctx->exec_counters, ctx->d_exec, sizeof(ctx->exec_counters), hipMemcpyDeviceToHost));
    if (o->collect_counters == 1) {
        unsigned long long c2[3 * C_N], c[C_N];
        CRT_HIP_CHECK(ctx, hipMemcpy(c2, ctx->d_counters, sizeof(c2), hipMemcpyDeviceToHost));
        for (int k = 0; k < C_N; k++) {
            c[k] = c2[k] + c2[C_N + k] + c2[2 * C_N + k];
            ctx->level_counters[k] = c2[k];
            ctx->shadow0_counters[k] = c2[C_N + k];
        }
        ctx->stats.box_tests = c[C_BOX]; ctx->stats.tri_tests = c[C_TRI]; ctx->stats.leaf_index_reads = c[C_LEAFIDX];
        ctx->stats.shaded_hits = c[C_HIT]; ctx->stats.light_evals = c[C_LIGHT]; ctx->stats.texel_fetches = c[C_TEXEL];
        ctx->stats.primary_rays = c[C_PRIMARY]; ctx->stats.secondary_rays = c[C_SECONDARY]; ctx->stats.shadow_rays = c[C_SHADOW];
    }
    return CRT_OK;
}

// The pixels a set of rectangles covers -- clamped to the image like renderRectangle does (RayTracer.cpp:84-85), overlaps
// merged -- as work items: one per 8x8 tile that has any covered pixel, with the mask of those pixels.  Returns the count
// of covered pixels.  out_tile = the tile itself; callers that pack their output renumber it.
static uint64_t coverage_items(uint32_t width, uint32_t height, const crt_rect *rects, uint32_t n_rects, std::vector<WorkItem> &items) {
    const uint32_t tx = (width + TILE - 1) / TILE, ty = (height + TILE - 1) / TILE;
    std::vector<uint64_t> masks((size_t)tx * ty, 0);
    for (uint32_t r = 0; r < n_rects; r++) {
        uint64_t row_lim = (uint64_t)rects[r].row + rects[r].height, col_lim = (uint64_t)rects[r].col + rects[r].width;
        if (row_lim > height) row_lim = height;
        if (col_lim > width) col_lim = width;
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
    items.clear();
    uint64_t pixels = 0;
    for (uint32_t t = 0; t < tx * ty; t++)
        if (masks[t]) {
            items.push_back(WorkItem{t, t, masks[t]});
            pixels += (uint64_t)__builtin_popcountll(masks[t]);
        }
    return pixels;
}

// first half of crt_render / crt_render_async: everything up to (not including) the wait for the device
static int render_enqueue(crt_ctx *ctx, const crt_options *o, const crt_rect *rects, uint32_t n_rects, float *out_rgb, uint8_t *out_rgb8) {
    if (!ctx) return CRT_ERR_INVALID;
    int rc = check_options(ctx, o);
    if (rc) return rc;
    if (n_rects && !rects) { ctx->error = "rects is NULL"; return CRT_ERR_INVALID; }
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // coverage: the union of the clamped rectangles (RayTracer.cpp:84-85), as 8x8 tiles with lane masks
    bool same = !ctx->cached_is_partition && ctx->cached_rects.size() == n_rects && n_rects > 0 &&
                memcmp(ctx->cached_rects.data(), rects, n_rects * sizeof(crt_rect)) == 0;
    if (!same) {
        std::vector<WorkItem> items;
        const uint64_t pixels = coverage_items(ctx->width, ctx->height, rects, n_rects, items);
        rc = ensure_items(ctx, items.size() ? items.size() : 1);
        if (rc) return rc;
        if (!items.empty())
            CRT_HIP_CHECK(ctx, hipMemcpy(ctx->d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        ctx->cached_rects.assign(rects, rects + n_rects);
        ctx->cached_is_partition = false;
        ctx->cached_n_items = (uint32_t)items.size();
        ctx->cached_pixels = pixels;
    }
    if (!ctx->ev_call0) {
        CRT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev_call0));
        CRT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev_call1));
    }
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_call0, ctx->stream));
    rc = launch_render(ctx, o, ctx->cached_n_items, ctx->d_frame, 0, ctx->stream);
    if (rc) return rc;
    const size_t values = (size_t)ctx->width * ctx->height * 3;
    if (out_rgb) CRT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgb, ctx->d_frame, values * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (out_rgb8) {  // PPMColor on the device (Color.cpp:12-16), a quarter of the bytes to copy
        rc = crt_quantize_device(ctx, ctx->d_frame, values, ctx->d_quant, ctx->stream);
        if (rc) return rc;
        CRT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgb8, ctx->d_quant, values, hipMemcpyDeviceToHost, ctx->stream));
    }
    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_call1, ctx->stream));
    ctx->pending = true;
    ctx->pending_options = *o;
    return CRT_OK;
}

// second half of crt_render / crt_wait: the frame enqueued by render_enqueue has to be finished before its statistics exist
static int render_finish(crt_ctx *ctx) {
    if (!ctx->pending) return CRT_OK;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    CRT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pending = false;
    float ms = 0;
    {
        const int slot = (int)((ctx->launches - 1) % crt_ctx::EV_RING);
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0[slot], ctx->ev4[slot]));
        ctx->stats.kernel_ms = ctx->cached_n_items ? ms : 0;
    }
    CRT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev_call0, ctx->ev_call1));
    ctx->stats.total_ms = ms;
    note_overflow(ctx);
    return fetch_counters(ctx, &ctx->pending_options, ctx->cached_pixels);
}

extern "C" int crt_render(crt_ctx *ctx, const crt_options *o, const crt_rect *rects, uint32_t n_rects, float *out_rgb) {
    int rc = render_enqueue(ctx, o, rects, n_rects, out_rgb, nullptr);
    if (rc) return rc;
    return render_finish(ctx);
}

extern "C" int crt_render_async(crt_ctx *ctx, const crt_options *o, const crt_rect *rects, uint32_t n_rects, float *out_rgb,
                                uint8_t *out_rgb8) {
    if (ctx && ctx->pending) {  // one frame per context at a time: finish the previous one first
        int rc = render_finish(ctx);
        if (rc) return rc;
    }
    return render_enqueue(ctx, o, rects, n_rects, out_rgb, out_rgb8);
}

// pinned host memory for crt_render_async's outputs (callers of the C ABI need no HIP headers for it)
extern "C" void *crt_alloc_pinned(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void crt_free_pinned(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int crt_wait(crt_ctx *ctx) {
    if (!ctx) return CRT_ERR_INVALID;
    return render_finish(ctx);
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
    crt_rect key{first, stride, 0u, 0u};
    bool same = ctx->cached_is_partition && ctx->cached_rects.size() == 1 && memcmp(ctx->cached_rects.data(), &key, sizeof(key)) == 0;
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
        ctx->cached_is_partition = true;
        ctx->cached_n_items = n;
        ctx->cached_pixels = pixels;
    }
    rc = launch_render(ctx, o, n, d_packed, 1, (hipStream_t)stream);
    if (rc) return rc;
    if (o->collect_counters) {  // counting renders are synchronous: the counters are read back here
        CRT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
        return fetch_counters(ctx, o, ctx->cached_pixels);
    }
    ctx->stats.pixels = ctx->cached_pixels;
    ctx->stats.counters_valid = 0;
    return CRT_OK;
}

extern "C" int crt_unpack_tiles_device(crt_ctx *ctx, const float *d_packed_all, uint32_t n_parts, uint64_t part_stride_floats,
                                       float *d_frame, void *stream) {
    if (!ctx || !d_packed_all || !d_frame || n_parts == 0) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n_tiles = ctx->tiles_x * ctx->tiles_y;
    // part p holds tiles p, p + n_parts, ...: the stride must cover the largest part, or the kernel would read past it
    const uint64_t tiles_per_part = ((uint64_t)n_tiles + n_parts - 1) / n_parts;
    if (part_stride_floats < tiles_per_part * 64 * 3) {
        ctx->error = "crt_unpack_tiles_device: part_stride_floats is smaller than one part's tiles (ceil(tiles / n_parts) * 192 floats)";
        return CRT_ERR_INVALID;
    }
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
    if (ctx->launches == 0) { ctx->error = "no render kernel has been launched"; return CRT_ERR_INVALID; }
    const int slot = (int)((ctx->launches - 1) % crt_ctx::EV_RING);
    CRT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev4[slot]));
    float f = 0;
    CRT_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->ev0[slot], ctx->ev4[slot]));
    *ms = f;
    ctx->stats.kernel_ms = f;
    return CRT_OK;
}

extern "C" int crt_kernel_times_ms(crt_ctx *ctx, double *out_phase_ms, uint32_t max_count, uint32_t *count) {
    if (!ctx || !count) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t n = ctx->launches < (uint64_t)crt_ctx::EV_RING ? ctx->launches : (uint64_t)crt_ctx::EV_RING;
    if (n > max_count) n = max_count;
    for (uint64_t i = 0; i < n; i++) {  // oldest first
        const int slot = (int)((ctx->launches - n + i) % crt_ctx::EV_RING);
        CRT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev4[slot]));
        float t = 0, a = 0, b = 0, c = 0, e = 0;
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&t, ctx->ev0[slot], ctx->ev4[slot]));
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&a, ctx->ev0[slot], ctx->ev1[slot]));
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&b, ctx->ev_s0[slot], ctx->ev_s1[slot]));
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&c, ctx->ev1[slot], ctx->ev2[slot]));
        CRT_HIP_CHECK(ctx, hipEventElapsedTime(&e, ctx->ev2[slot], ctx->ev3[slot]));
        if (out_phase_ms) {
            out_phase_ms[5 * i] = t; out_phase_ms[5 * i + 1] = a; out_phase_ms[5 * i + 2] = b;
            out_phase_ms[5 * i + 3] = c; out_phase_ms[5 * i + 4] = e;
        }
    }
    *count = (uint32_t)n;
    note_overflow(ctx);
    return CRT_OK;
}

extern "C" int crt_get_kernel_counters(crt_ctx *ctx, uint64_t closest[9], uint64_t shadow[9]) {
    if (!ctx || !closest || !shadow) return CRT_ERR_INVALID;
    for (int k = 0; k < C_PUBLIC; k++) { closest[k] = ctx->level_counters[k]; shadow[k] = ctx->shadow0_counters[k]; }
    return CRT_OK;
}

extern "C" int crt_get_executed_counters(crt_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return CRT_ERR_INVALID;
    out[0] = ctx->exec_counters[0] + ctx->exec_counters[2];  // the whole render
    out[1] = ctx->exec_counters[1] + ctx->exec_counters[3];
    out[2] = ctx->exec_counters[2];                          // of which shadow pass 0 (stream_trace_shadow_lean<0>)
    out[3] = ctx->exec_counters[3];
    return CRT_OK;
}

extern "C" int crt_get_executed_plan_tests(crt_ctx *ctx, uint64_t out[2]) {
    if (!ctx || !out) return CRT_ERR_INVALID;
    out[0] = ctx->exec_counters[4] + ctx->exec_counters[5];  // the whole render
    out[1] = ctx->exec_counters[5];                          // of which shadow pass 0
    return CRT_OK;
}

// diagnostics: the ray-stream pass's counters of the last frame (SC_* layout of kernel_stream.h)
extern "C" int crt_debug_stream_counts(crt_ctx *ctx, uint32_t *out, uint32_t max_words) {
    if (!ctx || !out) return CRT_ERR_INVALID;
    CRT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    CRT_HIP_CHECK(ctx, hipDeviceSynchronize());
    const uint32_t n = max_words < (uint32_t)SC_ALLOC_WORDS ? max_words : (uint32_t)SC_ALLOC_WORDS;
    CRT_HIP_CHECK(ctx, hipMemcpy(out, ctx->d_scounts, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CRT_OK;
}

// Which kernels a production frame of this context runs (bench.py names the roofline's kernel with it).
extern "C" int crt_describe_kernels(const crt_ctx *ctx, char *out, size_t size) {
    if (!ctx || !out || size == 0) return CRT_ERR_INVALID;
    const SceneArgs &A = ctx->scene;
    std::string d;
    if (ctx->mode == crt_ctx::MODE_LANES) d = "all=render_lanes<false>";
    else {
        const bool heavy = ctx->step_budget && A.nested_boxes && (A.top_fast || A.plan_seq);
        const bool lean = heavy && ctx->lean_ok && (A.plan_ok || A.plan_wide);
        const bool wide = lean && !A.plan_ok;
        const bool bvh = A.bvh_ok && ctx->tuning.bvh;
        if (bvh && ctx->tuning.level_queue) d = "level0=bvh_trace_level0<0>;shadow0=bvh_trace_shadow<0u, 0>;levels=bvh_trace_queue<0>";
        else if (bvh) d = "level0=bvh_trace_shade;shadow0=bvh_trace_shadow<0u, 0>;levels=bvh_trace_shade";
        else d = std::string("level0=") + (!lean ? "stream_trace_shade<false>" : wide ? "stream_trace_shade_plan_wide" : "stream_trace_shade_plan");
        if (!bvh) d += std::string(";shadow0=") + (!lean ? "stream_trace_shadow<false>" : wide ? "stream_trace_shadow_plan_wide<0u>" : "stream_trace_shadow_plan<0u>");
        if (!bvh) d += std::string(";levels=") + (!heavy ? "stream_trace_shade<false>" : "heavy_trace_closest");
    }
    d += std::string(";filter=") + (A.bvh_ok ? ctx->bvh_stats : "none (" + ctx->bvh_note + ")");
    snprintf(out, size, "%s", d.c_str());
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
    note_overflow(ctx);
    return CRT_OK;
}

// =================================================================================================
// crt_multi: one scene on several devices of a node, behind the same render call (SURVEY.md section 8b "multi-GPU handled
// inside the context", section 8e).  One context, one host thread and one stream per device; the covered 8x8 tiles are
// dealt round-robin over the devices; every device renders its tiles packed and copies them to device[0] over xGMI
// (hipMemcpyPeerAsync: point-to-point, no collective is needed inside one process); device[0] scatters them into its
// persistent colour buffer.  The same device may be listed more than once (several contexts on one GPU).
struct crt_multi {
    std::vector<crt_ctx *> ctx;
    std::vector<int> devices;
    std::vector<float *> d_packed;        // part p's packed tiles on its own device (part 0: inside d_staging)
    float *d_staging = nullptr;           // device[0]: every part's packed tiles, part after part
    size_t staging_tiles = 0;
    WorkItem *d_all_items = nullptr;      // device[0]: every covered tile {tile, slot in d_staging, mask}
    size_t all_items_cap = 0;
    std::vector<std::vector<WorkItem>> part_items;
    std::vector<uint32_t> slot_base;      // first slot of part p in d_staging
    std::vector<hipEvent_t> done;         // part p's tiles have arrived on device[0] (a staged part: in its pinned host buffer)
    // A part whose device cannot store into device[0]'s memory (hipDeviceCanAccessPeer says no, or hipDeviceEnablePeerAccess fails)
    // is STAGED: its tiles go to a pinned host buffer on its own stream and from there to device[0] on device[0]'s stream.
    // crt_multi_create says so (crt_multi_staged_parts, crt_multi_peer_note) instead of leaving the choice to the runtime.
    std::vector<uint8_t> staged;
    std::vector<float *> h_stage;
    std::vector<size_t> h_stage_tiles;
    std::string peer_note;
    bool force_staged = false;            // crt_debug_multi_force_staged: every part but the first staged (tests, one-GPU boxes)
    bool fail_next_alloc = false;         // crt_debug_multi_fail_next_alloc: the next re-partition fails at a part's buffer (tests)
    std::vector<crt_rect> cached_rects;
    uint32_t n_all_items = 0;
    uint64_t pixels = 0;
    std::string error;
    crt_stats stats{};
    // one worker thread per device: launching a frame is ~35 kernel launches per device, issued in parallel
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        bool go = false, quit = false, finished = false;
        int rc = CRT_OK;
    };
    std::vector<Worker *> workers;
    const crt_options *job_options = nullptr;
};

static int multi_part_launch(crt_multi *M, uint32_t p) {
    crt_ctx *ctx = M->ctx[p];
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->error = "hipSetDevice failed"; return CRT_ERR_HIP; }
    const uint32_t n = (uint32_t)M->part_items[p].size();
    int rc = launch_render(ctx, M->job_options, n, M->d_packed[p], 1, ctx->stream);
    if (rc) return rc;
    if (p != 0 && n) {  // part 0 renders straight into the staging buffer
        if (M->staged[p] || M->force_staged)
            CRT_HIP_CHECK(ctx, hipMemcpyAsync(M->h_stage[p], M->d_packed[p], (size_t)n * 192 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        else
            CRT_HIP_CHECK(ctx, hipMemcpyPeerAsync(M->d_staging + (size_t)M->slot_base[p] * 192, M->devices[0], M->d_packed[p],
                                                  M->devices[p], (size_t)n * 192 * sizeof(float), ctx->stream));
    }
    CRT_HIP_CHECK(ctx, hipEventRecord(M->done[p], ctx->stream));
    return CRT_OK;
}

static void multi_worker(crt_multi *M, uint32_t p) {
    crt_multi::Worker *W = M->workers[p];
    for (;;) {
        std::unique_lock<std::mutex> lock(W->m);
        W->cv.wait(lock, [&] { return W->go || W->quit; });
        if (W->quit) return;
        W->go = false;
        lock.unlock();
        const int rc = multi_part_launch(M, p);
        lock.lock();
        W->rc = rc;
        W->finished = true;
        W->cv.notify_all();
    }
}

extern "C" void crt_multi_destroy(crt_multi *M) {
    if (!M) return;
    for (crt_multi::Worker *W : M->workers) {
        if (!W) continue;
        { std::lock_guard<std::mutex> lock(W->m); W->quit = true; }
        W->cv.notify_all();
        if (W->th.joinable()) W->th.join();
        delete W;
    }
    for (size_t p = 0; p < M->ctx.size(); p++) {
        if (!M->ctx[p]) continue;
        (void)hipSetDevice(M->devices[p]);
        if (p < M->done.size() && M->done[p]) (void)hipEventDestroy(M->done[p]);
        if (p != 0 && p < M->d_packed.size() && M->d_packed[p]) (void)hipFree(M->d_packed[p]);
        if (p < M->h_stage.size() && M->h_stage[p]) (void)hipHostFree(M->h_stage[p]);
    }
    if (!M->devices.empty()) {
        (void)hipSetDevice(M->devices[0]);
        if (M->d_staging) (void)hipFree(M->d_staging);
        if (M->d_all_items) (void)hipFree(M->d_all_items);
    }
    for (crt_ctx *c : M->ctx) crt_destroy(c);
    delete M;
}

extern "C" int crt_multi_create(const crt_scene_desc *scene, const int *devices, uint32_t n_devices, const crt_tuning *tuning,
                                crt_multi **out) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) { g_create_error = "crt_multi_create: 1..64 devices"; return CRT_ERR_INVALID; }
    crt_multi *M = new (std::nothrow) crt_multi();
    if (!M) return CRT_ERR_NOMEM;
    M->devices.assign(devices, devices + n_devices);
    M->ctx.assign(n_devices, nullptr);
    M->d_packed.assign(n_devices, nullptr);
    M->done.assign(n_devices, nullptr);
    M->staged.assign(n_devices, 0);
    M->h_stage.assign(n_devices, nullptr);
    M->h_stage_tiles.assign(n_devices, 0);
    M->part_items.resize(n_devices);
    M->slot_base.assign(n_devices, 0);
    for (uint32_t p = 0; p < n_devices; p++) {
        int rc = crt_create_tuned(scene, devices[p], tuning, &M->ctx[p]);
        if (rc != CRT_OK) { crt_multi_destroy(M); return rc; }
        if (hipSetDevice(devices[p]) != hipSuccess || hipEventCreateWithFlags(&M->done[p], hipEventDisableTiming) != hipSuccess) {
            g_create_error = "crt_multi_create: event creation failed";
            crt_multi_destroy(M);
            return CRT_ERR_HIP;
        }
        // direct xGMI stores to device[0] where the platform allows them; where it does not, say so and stage that part's tiles
        if (devices[p] != devices[0]) {
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, devices[p], devices[0]);
            const char *what = "hipDeviceCanAccessPeer";
            if (e == hipSuccess && can) {
                what = "hipDeviceEnablePeerAccess";
                e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
            }
            (void)hipGetLastError();
            if (e != hipSuccess || !can) {
                M->staged[p] = 1;
                M->peer_note += "part " + std::to_string(p) + " (device " + std::to_string(devices[p]) + " -> " + std::to_string(devices[0]) + "): " +
                                (e != hipSuccess ? std::string(what) + " failed: " + hipGetErrorString(e) : std::string("hipDeviceCanAccessPeer: no peer access")) +
                                "; its tiles are staged through pinned host memory\n";
            }
        }
    }
    M->workers.assign(n_devices, nullptr);
    for (uint32_t p = 0; p < n_devices; p++) {
        M->workers[p] = new crt_multi::Worker();
        M->workers[p]->th = std::thread(multi_worker, M, p);
    }
    *out = M;
    return CRT_OK;
}

extern "C" const char *crt_multi_last_error(const crt_multi *M) { return M ? M->error.c_str() : g_create_error.c_str(); }
extern "C" uint32_t crt_multi_device_count(const crt_multi *M) { return M ? (uint32_t)M->ctx.size() : 0u; }
extern "C" crt_ctx *crt_multi_context(crt_multi *M, uint32_t part) { return (M && part < M->ctx.size()) ? M->ctx[part] : nullptr; }

extern "C" int crt_multi_set_camera(crt_multi *M, const float position[3], const float matrix[9]) {
    if (!M) return CRT_ERR_INVALID;
    for (crt_ctx *c : M->ctx) {
        int rc = crt_set_camera(c, position, matrix);
        if (rc) return rc;
    }
    return CRT_OK;
}

extern "C" int crt_multi_render(crt_multi *M, const crt_options *o, const crt_rect *rects, uint32_t n_rects, float *out_rgb) {
    if (!M) return CRT_ERR_INVALID;
    crt_ctx *c0 = M->ctx[0];
    auto fail = [&](crt_ctx *c, int rc) { M->error = c->error; return rc; };
    int rc = check_options(c0, o);
    if (rc) return fail(c0, rc);
    if (n_rects && !rects) { M->error = "rects is NULL"; return CRT_ERR_INVALID; }
    const uint32_t N = (uint32_t)M->ctx.size();
    const bool same = M->cached_rects.size() == n_rects && n_rects > 0 &&
                      memcmp(M->cached_rects.data(), rects, n_rects * sizeof(crt_rect)) == 0;
    if (!same) {
        // (a rebuild that fails half-way must be redone by the next call, not skipped because the rectangles look familiar)
        M->cached_rects.clear();
        M->n_all_items = 0;
        // the covered tiles, dealt round-robin: neighbouring tiles cost about the same, so every device gets every kind
        std::vector<WorkItem> items;
        M->pixels = coverage_items(c0->width, c0->height, rects, n_rects, items);
        for (uint32_t p = 0; p < N; p++) M->part_items[p].clear();
        for (size_t j = 0; j < items.size(); j++) {
            std::vector<WorkItem> &part = M->part_items[j % N];
            part.push_back(WorkItem{items[j].tile, (uint32_t)part.size(), items[j].mask});
        }
        std::vector<WorkItem> all;
        uint32_t slot = 0;
        for (uint32_t p = 0; p < N; p++) {
            M->slot_base[p] = slot;
            for (const WorkItem &w : M->part_items[p]) all.push_back(WorkItem{w.tile, slot + w.out_tile, w.mask});
            slot += (uint32_t)M->part_items[p].size();
        }
        M->n_all_items = (uint32_t)all.size();
        CRT_HIP_CHECK(c0, hipSetDevice(M->devices[0]));
        if (all.size() > M->all_items_cap) {
            if (M->d_all_items) (void)hipFree(M->d_all_items);
            M->d_all_items = nullptr;
            M->all_items_cap = 0;
            if (hipMalloc((void **)&M->d_all_items, all.size() * sizeof(WorkItem)) != hipSuccess) { M->error = "out of device memory"; return CRT_ERR_NOMEM; }
            M->all_items_cap = all.size();
        }
        if (all.size() > M->staging_tiles) {
            if (M->d_staging) (void)hipFree(M->d_staging);
            M->d_staging = nullptr;
            M->staging_tiles = 0;
            if (hipMalloc((void **)&M->d_staging, all.size() * 192 * sizeof(float)) != hipSuccess) { M->error = "out of device memory"; return CRT_ERR_NOMEM; }
            M->staging_tiles = all.size();
        }
        if (!all.empty()) CRT_HIP_CHECK(c0, hipMemcpy(M->d_all_items, all.data(), all.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        for (uint32_t p = 0; p < N; p++) {
            crt_ctx *c = M->ctx[p];
            const std::vector<WorkItem> &part = M->part_items[p];
            if (hipSetDevice(M->devices[p]) != hipSuccess) { M->error = "hipSetDevice failed"; return CRT_ERR_HIP; }
            rc = ensure_items(c, part.size() ? part.size() : 1);
            if (rc) return fail(c, rc);
            if (!part.empty() && hipMemcpy(c->d_items, part.data(), part.size() * sizeof(WorkItem), hipMemcpyHostToDevice) != hipSuccess) {
                M->error = "item upload failed";
                return CRT_ERR_HIP;
            }
            c->cached_rects.clear();  // the context's own crt_render cache no longer describes d_items
            c->cached_is_partition = false;
            if (p != 0) {
                if (M->d_packed[p]) (void)hipFree(M->d_packed[p]);
                M->d_packed[p] = nullptr;
                if (M->fail_next_alloc || hipMalloc((void **)&M->d_packed[p], (part.size() ? part.size() : 1) * 192 * sizeof(float)) != hipSuccess) {
                    M->fail_next_alloc = false;  // (crt_debug_multi_fail_next_alloc: the rebuild stops half-way, as a failed allocation would leave it)
                    M->error = "out of device memory";
                    return CRT_ERR_NOMEM;
                }
                if ((M->staged[p] || M->force_staged) && part.size() > M->h_stage_tiles[p]) {
                    if (M->h_stage[p]) (void)hipHostFree(M->h_stage[p]);
                    M->h_stage[p] = nullptr;
                    M->h_stage_tiles[p] = 0;
                    if (hipHostMalloc((void **)&M->h_stage[p], part.size() * 192 * sizeof(float), hipHostMallocPortable) != hipSuccess) { M->error = "out of pinned host memory"; return CRT_ERR_NOMEM; }
                    M->h_stage_tiles[p] = part.size();
                }
            }
        }
        M->cached_rects.assign(rects, rects + n_rects);
    }
    M->d_packed[0] = M->d_staging + (size_t)M->slot_base[0] * 192;
    // launch every part from its own thread, then gather on device[0]
    M->job_options = o;
    const auto wall0 = std::chrono::steady_clock::now();
    for (uint32_t p = 0; p < N; p++) {
        crt_multi::Worker *W = M->workers[p];
        { std::lock_guard<std::mutex> lock(W->m); W->finished = false; W->go = true; }
        W->cv.notify_all();
    }
    int first_rc = CRT_OK;
    for (uint32_t p = 0; p < N; p++) {
        crt_multi::Worker *W = M->workers[p];
        std::unique_lock<std::mutex> lock(W->m);
        W->cv.wait(lock, [&] { return W->finished; });
        if (W->rc != CRT_OK && first_rc == CRT_OK) { first_rc = W->rc; M->error = M->ctx[p]->error; }
    }
    if (first_rc != CRT_OK) {
        for (uint32_t p = 0; p < N; p++) { (void)hipSetDevice(M->devices[p]); (void)hipDeviceSynchronize(); }
        return first_rc;
    }
    CRT_HIP_CHECK(c0, hipSetDevice(M->devices[0]));
    for (uint32_t p = 1; p < N; p++) {
        CRT_HIP_CHECK(c0, hipStreamWaitEvent(c0->stream, M->done[p], 0));
        if ((M->staged[p] || M->force_staged) && !M->part_items[p].empty())  // the second leg of a staged part: pinned host -> device[0]
            CRT_HIP_CHECK(c0, hipMemcpyAsync(M->d_staging + (size_t)M->slot_base[p] * 192, M->h_stage[p],
                                             M->part_items[p].size() * 192 * sizeof(float), hipMemcpyHostToDevice, c0->stream));
    }
    if (M->n_all_items) {
        const uint64_t threads = (uint64_t)M->n_all_items * 64;
        hipLaunchKernelGGL(unpack_items_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c0->stream, M->d_staging,
                           M->d_all_items, M->n_all_items, c0->d_frame, c0->width, c0->height, c0->tiles_x);
        CRT_HIP_CHECK(c0, hipGetLastError());
    }
    if (out_rgb)
        CRT_HIP_CHECK(c0, hipMemcpyAsync(out_rgb, c0->d_frame, (size_t)c0->width * c0->height * 3 * sizeof(float), hipMemcpyDeviceToHost, c0->stream));
    CRT_HIP_CHECK(c0, hipStreamSynchronize(c0->stream));
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    // statistics: counters summed over the parts, device time = the slowest part's
    crt_stats total{};
    for (uint32_t p = 0; p < N; p++) {
        crt_ctx *c = M->ctx[p];
        if (hipSetDevice(M->devices[p]) != hipSuccess) continue;
        (void)hipStreamSynchronize(c->stream);
        uint64_t px = 0;
        for (const WorkItem &w : M->part_items[p]) px += (uint64_t)__builtin_popcountll(w.mask);
        rc = fetch_counters(c, o, px);
        if (rc) return fail(c, rc);
        note_overflow(c);
        double ms = 0;
        if (!M->part_items[p].empty() && crt_kernel_elapsed_ms(c, &ms) == CRT_OK && ms > total.kernel_ms) total.kernel_ms = ms;
        total.box_tests += c->stats.box_tests; total.tri_tests += c->stats.tri_tests; total.leaf_index_reads += c->stats.leaf_index_reads;
        total.shaded_hits += c->stats.shaded_hits; total.light_evals += c->stats.light_evals; total.texel_fetches += c->stats.texel_fetches;
        total.primary_rays += c->stats.primary_rays; total.secondary_rays += c->stats.secondary_rays; total.shadow_rays += c->stats.shadow_rays;
        total.fallback_frames += c->stats.fallback_frames;
        total.queue_bytes += c->stats.queue_bytes;
        total.queue_regrows += c->stats.queue_regrows;
    }
    total.pixels = M->pixels;
    total.total_ms = wall_ms;
    total.counters_valid = o->collect_counters == 1 ? 1u : 0u;
    M->stats = total;
    return CRT_OK;
}

extern "C" uint32_t crt_multi_staged_parts(const crt_multi *M) {
    uint32_t n = 0;
    if (M) for (size_t p = 1; p < M->staged.size(); p++) n += (M->staged[p] || M->force_staged) ? 1u : 0u;
    return n;
}

extern "C" const char *crt_multi_peer_note(const crt_multi *M) { return M ? M->peer_note.c_str() : ""; }

extern "C" int crt_debug_multi_force_staged(crt_multi *M, int on) {
    if (!M) return CRT_ERR_INVALID;
    M->force_staged = on != 0;
    M->cached_rects.clear();  // the next render allocates the pinned buffers
    return CRT_OK;
}

extern "C" int crt_debug_multi_fail_next_alloc(crt_multi *M) {
    if (!M) return CRT_ERR_INVALID;
    M->fail_next_alloc = true;
    return CRT_OK;
}

extern "C" int crt_multi_get_stats(crt_multi *M, crt_stats *out) {
    if (!M || !out) return CRT_ERR_INVALID;
    *out = M->stats;
    return CRT_OK;
}

extern "C" int crt_multi_read_quantized(crt_multi *M, uint8_t *out_rgb8) {
    if (!M) return CRT_ERR_INVALID;
    int rc = crt_read_quantized(M->ctx[0], out_rgb8);
    if (rc) M->error = M->ctx[0]->error;
    return rc;
}
