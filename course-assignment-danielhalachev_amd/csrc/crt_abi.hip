// crt_abi.hip -- the render entry points of include/crt_hip.h: coverage (the reference's bucket rectangles as 8x8 work items), crt_render /
// crt_render_async / crt_wait, the device-resident tile calls of the multi-GPU partition, the quantiser, statistics and diagnostics.
// replaces the bucket schedulers + renderRectangle (RayTracer.cpp:141-158, 82-112) and PPMColor (Color.cpp:12-16).
#include "crt_internal.h"

namespace {


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

// PPMColor (Color.cpp:12-16): (unsigned short)(std::clamp(c, 0.0f, 1.0f) * 255)
__global__ void quantize_kernel(const float *rgb, uint64_t n, uint8_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = rgb[i];
    const float cl = (c < 0.0f) ? 0.0f : ((1.0f < c) ? 1.0f : c);
    out[i] = (uint8_t)(unsigned short)(cl * 255);
}

}  // namespace
void launch_unpack_items(const float *packed, const WorkItem *items, uint32_t n_items, float *frame, uint32_t width, uint32_t height,
                         uint32_t tiles_x, hipStream_t stream) {
    hipLaunchKernelGGL(unpack_items_kernel, dim3((uint32_t)(((uint64_t)n_items * 64 + 255) / 256)), dim3(256), 0, stream, packed, items, n_items, frame, width, height, tiles_x);
}

uint64_t coverage_items(uint32_t width, uint32_t height, const crt_rect *rects, uint32_t n_rects, std::vector<WorkItem> &items) {
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
