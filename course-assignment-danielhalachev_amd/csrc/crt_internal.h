// crt_internal.h -- what the translation units behind include/crt_hip.h share: the context, the layouts the kernels and the host
// agree on (kernel_common.h), and the few functions that cross a file boundary.  Nothing here is part of the ABI.
//   crt_scene.hip     crt_create / crt_destroy: the flattened scene and everything derived from it, uploaded once
//   crt_launch.hip    the kernels and one frame's launches: frame plan, queue sizing, events
//   crt_abi.hip       the render entry points, tiles, statistics
//   crt_multi.hip     one scene on several devices behind one call
//   crt_testhooks.hip unit-test hooks (libcrt_hip_test.so only)
#pragma once

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
#include "kernel_common.h"

#define CRT_INTERNAL __attribute__((visibility("hidden")))

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
    hipStream_t early = nullptr;      // the level queue's launch starts WITH level 0 on this one (crt_launch.hip)
    hipEvent_t ev_reset[EV_RING] = {}, ev_queue[EV_RING] = {};   // the frame's counters are zeroed / the level queue's launch has ended
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

extern CRT_INTERNAL std::string g_create_error;   // crt_scene.hip

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

// ---- crt_launch.hip
CRT_INTERNAL int check_options(crt_ctx *ctx, const crt_options *o);
CRT_INTERNAL int ensure_items(crt_ctx *ctx, size_t n);
CRT_INTERNAL int launch_render(crt_ctx *ctx, const crt_options *o, uint32_t n_items, float *d_out, uint32_t packed, hipStream_t stream);
CRT_INTERNAL void note_overflow(crt_ctx *ctx);
CRT_INTERNAL int fetch_counters(crt_ctx *ctx, const crt_options *o, uint64_t pixels);
// ---- crt_abi.hip
CRT_INTERNAL uint64_t coverage_items(uint32_t width, uint32_t height, const crt_rect *rects, uint32_t n_rects, std::vector<WorkItem> &items);
CRT_INTERNAL void launch_unpack_items(const float *packed, const WorkItem *items, uint32_t n_items, float *frame, uint32_t width, uint32_t height,
                                      uint32_t tiles_x, hipStream_t stream);
