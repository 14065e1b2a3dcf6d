// crt_multi.h -- the several-devices context (crt_multi.hip; crt_testhooks.hip switches two of its flags)
#pragma once
#include "crt_internal.h"

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
