// crt_multi.hip -- one scene on several devices of a node behind one render call (include/crt_hip.h: crt_multi_*).
#include "crt_internal.h"
#include "crt_multi.h"

// =================================================================================================
// crt_multi: one scene on several devices of a node, behind the same render call (SURVEY.md section 8b "multi-GPU handled
// inside the context", section 8e).  One context, one host thread and one stream per device; the covered 8x8 tiles are
// dealt round-robin over the devices; every device renders its tiles packed and copies them to device[0] over xGMI
// (hipMemcpyPeerAsync: point-to-point, no collective is needed inside one process); device[0] scatters them into its
// persistent colour buffer.  The same device may be listed more than once (several contexts on one GPU).

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
        launch_unpack_items(M->d_staging, M->d_all_items, M->n_all_items, c0->d_frame, c0->width, c0->height, c0->tiles_x, c0->stream);
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
