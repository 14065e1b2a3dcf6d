// crt_launch.hip -- the kernels of the MI355X (gfx950) implementation behind include/crt_hip.h, and one frame's launches.
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
#include "crt_internal.h"
#include "glibc_powf.h"

namespace {

#include "kernel_lane.h"
#include "kernel_stream.h"
#include "kernel_heavy.h"
#include "kernel_plan.h"
#include "kernel_bvh.h"

}  // namespace

int check_options(crt_ctx *ctx, const crt_options *o) {
    if (!o) { ctx->error = "options is NULL"; return CRT_ERR_INVALID; }
    if (o->use_gi && o->collect_counters == 2) { ctx->error = "collect_counters == 2 belongs to the ray-stream kernels; the GI mode renders with render_lanes"; return CRT_ERR_INVALID; }
    if (o->use_gi && (o->gi_sample_size > 64u || o->rays_per_pixel > 65536u)) { ctx->error = "gi_sample_size > 64 or rays_per_pixel > 65536"; return CRT_ERR_INVALID; }
    if (o->max_depth > 4096) { ctx->error = "max_depth too large"; return CRT_ERR_INVALID; }
    return CRT_OK;
}

int ensure_items(crt_ctx *ctx, size_t n) {
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
    uint32_t widest;              // rays of the widest level below level 0 a frame ago (0: not known)
    uint32_t level_budget;        // steps after which a deeper level's per-lane walk is evicted
    const uint32_t *prev;         // counters of a completed frame of this size and kind, or null
    bool last_resort;             // render_lanes behind the stream pass
};

// Levels 0 .. MAX_DEPTH on `stream`; after level 0 the bulk shadow pass (and the walks it gives up) on the side stream.
static int launch_stream_levels(crt_ctx *ctx, const crt_options *o, FramePlan &P, KernelArgs &A, hipStream_t stream) {
    // the frame's counters, the level queue's words, the tallies: zeroed (and the shadow queue's fill preset to level 0's fixed slots) in one launch
    hipLaunchKernelGGL(stream_frame_reset, dim3(1), dim3(256), 0, stream, ctx->d_scounts, P.queue ? ctx->d_lq_words : (uint32_t *)nullptr,
                       P.exec_count ? ctx->d_exec : (unsigned long long *)nullptr,
                       ctx->frame.fixed0 ? (uint32_t)P.vitems * 64u * ctx->n_lights : 0u);
    const uint32_t side_per_cu = ctx->tuning.side_blocks;  // workgroups per CU of the bulk shadow pass beside the levels
    A.exec_count = P.exec_count ? 1u : 0u;
    A.exec_counters = ctx->d_exec;
    A.exec_plan = ctx->d_exec + 4;
    // the plan kernels pay a wave-uniform loop per refill, whatever the number of new rays: refill in bundles
    A.bundle = REFILL_BUNDLE;
    A.wave_prio = 3u;  // the levels' waves (the frame's critical path) ahead of the bulk shadow pass's, which share their SIMDs
    A.force_whole = 0u;
    const uint32_t plds = ctx->scene.plan_list_words * BLOCK * (uint32_t)sizeof(uint32_t);  // kernel_plan.h: mesh lists
    A.chunk = ctx->tuning.fetch_chunk >> 16;   // (level 0's claims; crt_tuning::fetch_chunk)
    KernelArgs S = A;  // argument block of the bulk shadow pass
    S.wave_prio = 0u;
    S.chunk = ctx->tuning.fetch_chunk & 0xFFFFu;
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
    // every level below level 0: one launch that feeds itself (kernel_bvh.h); one workgroup per CU holds more lanes than the widest
    // level of a frame of this size has rays
    const bool early_queue = P.queue && side_per_cu && !(ctx->tuning.level_queue & 512u) && !g_debug_sync;   // (level_queue bit 9: the launch behind level 0, as it used to be)
    auto launch_queue = [&](hipStream_t where) {
        KernelArgs Q = A;
        const uint32_t qblocks = std::min(P.lane_blocks, (uint32_t)ctx->num_cus * std::max(1u, ctx->tuning.level_queue & 15u));
        // A reflection child continues in the lane that shaded its parent (no hand-over through memory) where the launch is a matter of
        // latency -- few rays: HW14 3.37 -> 3.29 ms, an eighth of its tiles 2.25 -> 2.04 -- and goes through the queue like the others where
        // it is a matter of balance (HW11, 65 - 85 k rays per level: 4.0 ms through the queue, 4.5 with chains kept in their lanes)
        Q.force_whole = (P.widest > 70000u || (ctx->tuning.level_queue & 256u)) ? 1u : 0u;
        Q.bundle = ctx->tuning.level0_budget ? ctx->tuning.level0_budget - 1u : 15u;   // (development: turns between two housekeeping rounds, as a mask)
        if (P.exec_count) launch(bvh_trace_queue<BVH_TALLY>, qblocks, where, Q);
        else if (ctx->tuning.bvh == 2) launch(bvh_trace_queue<BVH_CHECKED>, qblocks, where, Q);
        else launch(bvh_trace_queue<BVH_PLAIN>, qblocks, where, Q);
    };
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
            // The level queue's launch starts WITH level 0, on a stream of its own: its waves sleep until level 0's first reflective or
            // refractive hits reserve entries, and a pixel's chain of up to eight dependent walks -- the frame's critical path -- begins while
            // the other primary rays are still being walked, not after the last of them.  It may end once level 0 has (LQ_LEVEL0).
            if (early_queue && o->max_depth >= 1) {
                CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_reset[P.slot], stream));
                CRT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->early, ctx->ev_reset[P.slot], 0));
                launch_queue(ctx->early);
                if (ctx->tuning.level_queue & 4096u) {
                    // (tests: level 0 held back until that launch has ENDED -- what a profiler that serialises launches, or two streams on one
                    //  hardware queue, do to it: the launch must sit out its patience and leave, and the one behind level 0 must walk everything)
                    CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_queue[P.slot], ctx->early));
                    CRT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_queue[P.slot], 0));
                }
            }
            if (P.exec_count) launch(bvh_trace_level0<BVH_TALLY>, P.lane_blocks, stream, A);
            else if (ctx->tuning.bvh == 2) launch(bvh_trace_level0<BVH_CHECKED>, P.lane_blocks, stream, A);
            else launch(bvh_trace_level0<BVH_PLAIN>, P.lane_blocks, stream, A);
        }
        else if (P.bvh && P.gi) {   // the GI mode: the walk alone, then the level's shading with every lane busy (as the plan kernels' GI builds)
            launch(bvh_trace_shade_gi, P.lane_blocks, stream, A, g);
            launch(stream_shade_all<true>, P.lane_blocks, stream, A, g);
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
            if (early_queue && o->max_depth >= 1) {
                // ... and once more behind level 0 AND the first launch: nothing promises that two streams' kernels run side by side, so the
                // first launch's patience with an empty queue is bounded, and what it left is walked here (normally nothing: ~15 us)
                CRT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->early, ctx->ev_fork[P.slot], 0));
                launch_queue(ctx->early);
                CRT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_queue[P.slot], ctx->early));
                CRT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_queue[P.slot], 0));
            }
            else if (o->max_depth >= 1) launch_queue(stream);
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
    if (P.queue) {}   // (the level queue's lanes have walked the deeper levels' shadow rays themselves: kernel_bvh.h, BVH_SHADOWS)
    else if (P.bvh) {
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
    // (the GI mode too -- level by level, its levels are millions of rays -- except in the tallying and bounds-checked builds, which have no GI kernels)
    const bool bvh = stream_mode && SC.bvh_ok && ctx->tuning.bvh && !count && !(gi && (exec_count || ctx->tuning.bvh == 2));
    // ... the levels below level 0 as one self-feeding launch -- once a frame of this size is known to fit its queues: an attempt that
    // PROBES them (launch_render) runs level by level, because a level-by-level run that overflows leaves the counts the next attempt is
    // sized from, where the level queue stops in the middle of all its levels at once (same pixels either way)
    // And only where it pays: the queue trades launches and their tails (latency) for hand-overs through memory (throughput).  A frame whose
    // levels hold hundreds of thousands of rays each is throughput -- a mirror room at 1080p: 11.5 ms level by level, 20 ms through the queue;
    // HW12 at 3840x2160 (229 k rays per level): the same either way; HW14 / HW11 (20 - 85 k): 4.1 -> 3.3, 5.2 -> 4.0 ms -- so the widest
    // level of the last completed frame of this size decides.
    uint32_t widest = 0;
    if (ctx->last_counts_items == vitems && ctx->last_counts_cfg == frame_config_of(o))
        for (uint32_t g = 1; g <= o->max_depth && g < (uint32_t)MAX_GENERATIONS; g++) widest = std::max(widest, ctx->last_counts[SC_COUNT + g]);
    const bool queue = bvh && !gi && ctx->tuning.level_queue && last_resort && widest <= 250000u;
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
    FramePlan P{slot, n_items, vitems, lane_blocks, gi, count, exec_count, heavy, lean, wide, bvh, queue, widest, 0u, nullptr, last_resort};
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
int launch_render(crt_ctx *ctx, const crt_options *o, uint32_t n_items, float *d_out, uint32_t packed, hipStream_t stream) {
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
void note_overflow(crt_ctx *ctx) {
    harvest_counts(ctx);
    ctx->overflows = ctx->fallbacks_seen;
    ctx->stats.fallback_frames = (uint32_t)ctx->overflows;
    ctx->stats.queue_bytes = ctx->queue_bytes;
    ctx->stats.queue_regrows = ctx->regrows;
}

int fetch_counters(crt_ctx *ctx, const crt_options *o, uint64_t pixels) {
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