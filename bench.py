#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s on the HW14 'Optimizations 02'-like scene, 1920x1080, depth 8.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full frame of the per-pixel hot path.  The scene and its flattened tree are resident in
HBM before the timed region.  With N > 1 the frame's 8x8 tiles are dealt round-robin over the ranks (one
process per GPU, scene replicated), each rank renders its tiles into a packed device buffer and one RCCL
gather brings them to rank 0, which scatters them into the row-major frame.  The timed region
is bracketed by barrier + torch.cuda.synchronize() on both sides, the MAX over ranks is taken, and rank 0
prints ONE JSON line.  `value` is the whole job's Mpixels/s as SURVEY.md section 8(d) defines the metric -- the
reference's timing window ends with the float frame in HOST memory (RayTracer.cpp:207,289-293) -- so every timed
frame is copied to pinned host memory inside the timed region, on a copy stream, frame k's copy beside frame
k + 1's kernels.  `value_device_resident` is the same loop without the copies.

PyTorch is plumbing here (device buffers, streams, torch.distributed); the renderer is libcrt_hip.so.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_MEASURED_GBS = 6290.0  # ... and the streaming rate measured there (float4 copy, 79 %)
N_CUS, N_SIMDS = 256, 1024
WORKLOAD = "hw14"


def algorithmic_bytes(c, width, height, textured=False):
    """B_alg = 32 N_box + 48 N_tri + 4 N_leafidx + 72 N_hit (+36 textured) + 16 N_light + 12 N_texel
    + 12 W H  (bytes per frame)."""
    return (32 * c["box_tests"] + 48 * c["tri_tests"] + 4 * c["leaf_index_reads"]
            + (72 + (36 if textured else 0)) * c["shaded_hits"] + 16 * c["light_evals"] + 12 * c["texel_fetches"]
            + 12 * width * height)


def cpu_baseline(sc, scene_name, depth, w, h, budget_s=30.0, gpu_frame=None):
    """Times the REAL reference (oracle/_ref, built from /root/reference in the dev container and shipped as
    a binary) -- or, if that binary is absent, the oracle's C restatement -- on the same workload at the same
    resolution (one frame; about 9 s for the headline configuration on the GPU box's host cores).  A frame that
    would take longer than the budget is sampled at a quarter of the width and height instead, and says so.
    The frame the reference renders here is not thrown away: `gpu_frame` (the last frame the timed region delivered to host
    memory) is compared with it float for float -- the reference as CHECKER of the whole full-size frame, in this very run
    (`frame_matches_gpu`, `differing_floats`; null when the baseline is a sample of another size)."""
    from oracle import oracle_api as oa
    full = (w, h)
    if w * h > 2500000:  # HW12 at 3840x2160 would take minutes: sample it
        w, h = w // 4, h // 4
    scene = sc.make(scene_name, width=w, height=h)
    blob = sc.to_blob(scene)
    cores = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
        cores = min(cores, affinity)
    except Exception:
        pass
    buckets = int(scene["settings"]["image_settings"]["bucket_size"])
    size_note = "the full %dx%d frame" % (w, h) if (w, h) == full else "%dx%d (a 1/16 sample of %dx%d)" % (w, h, full[0], full[1])
    if oa.reference_available(textured=bool(scene.get("textures"))):
        ref_frame, info = oa.reference_render(blob, max_depth=depth)
        rep = 1
        if info["render_s"] * 3 < budget_s:
            rep = max(1, min(8, int(budget_s / max(info["render_s"], 1e-3)) - 1))
            ref_frame, info = oa.reference_render(blob, max_depth=depth, repeat=rep)
        secs = info["render_s"]
        matches, differing = None, None
        if gpu_frame is not None and (w, h) == full:
            import numpy as np
            g = np.ascontiguousarray(gpu_frame, dtype=np.float32).reshape(h, w, 3)
            diff = (g.view(np.uint32) != ref_frame.view(np.uint32)) & ~(np.isnan(g) & np.isnan(ref_frame))
            differing = int(diff.sum())
            matches = differing == 0
        # the reference's thread pool has hardware_concurrency threads, but only one job per bucket (RayTracer.cpp:141-158)
        busy = min(int(info["threads"]), buckets)
        return {"value": round(w * h / secs / 1e6, 5), "unit": "Mpixels/s", "cores": busy,
                "kind": "reference", "host_threads": int(info["threads"]), "seconds": round(secs, 4),
                "frame_matches_gpu": matches, "differing_floats": differing,
                "sample": "same %s-like scene (%d triangles), %s, depth %d: the reference's own RayTracer::render "
                          "(oracle/_ref, BVHBucketsThreadPool mode: %d buckets, so at most %d busy threads of the %d the "
                          "pool starts), best of %d"
                          % (scene_name.upper(), sc.triangle_count(scene), size_note, depth, buckets, busy,
                             int(info["threads"]), rep)}
    o = oa.OracleScene(blob)
    t0 = time.time()
    o.render(depth, threads=cores)
    secs = time.time() - t0
    return {"value": round(w * h / secs / 1e6, 5), "unit": "Mpixels/s", "cores": min(cores, buckets), "kind": "port",
            "seconds": round(secs, 4),
            "sample": "same %s-like scene, %s, depth %d, oracle/cpu_ref.c with OpenMP over the %d buckets"
                      % (scene_name.upper(), size_note, depth, buckets)}


def committed_evidence(profiles_dir, key, running_hash):
    """The counters committed under profiles/ -- used ONLY when they were collected from the kernel sources that are running.
    Returns (pmc, traffic, why_not): the parsed r04_pmc.json or None; the dominant kernel's HBM bytes per launch
    (hbm_traffic.json[key]) or None; and, when the PMC file is not used, the reason (a string for roofline.evidence)."""
    traffic = None
    try:
        with open(os.path.join(profiles_dir, "hbm_traffic.json")) as f:
            tj = json.load(f)
        if key in tj and tj[key].get("csrc_sha256") == running_hash:   # (counters of THESE kernels only)
            traffic = tj[key]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    try:
        with open(os.path.join(profiles_dir, "r04_pmc.json")) as f:
            pmc = json.load(f)
    except Exception as e:
        return None, None, "profiles/r04_pmc.json not usable: %r" % (e,)
    if pmc.get("csrc_sha256") != running_hash:
        return None, None, "profiles/r04_pmc.json was collected from other kernel sources (csrc sha256 %s..., running %s...): not used" % (
            str(pmc.get("csrc_sha256"))[:12], running_hash[:12])
    return pmc, traffic, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default=WORKLOAD)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="frames in flight of the extra `pipelined` measurement (default 3, 0 = skipped); `value` is always one frame at a time")
    ap.add_argument("--no-alone", action="store_true", help="skip roofline.alone (keeps a rocprofv3 --stats run of this command to the in-frame launches)")
    ap.add_argument("--tuning", default="", help="development: crt_tuning fields as 'name=value ...' (default: the library's defaults)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the collective path (process group, gather to rank 0) even with one rank: exercises the RCCL "
                         "branch on a one-GPU box (tests/test_gpu_cli.py)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="development only: N processes share GPU 0 and gather through gloo on host buffers "
                         "(exercises the multi-rank code path on a one-GPU box; its numbers mean nothing)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist   # the gather is a collective (with one rank: a copy, through the same RCCL call)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    pkg = importlib.import_module("course-assignment-danielhalachev_amd")
    sc = pkg.scenes
    gen, w0, h0, d0 = sc.CONFIGS[args.scene]
    W, H, depth = args.width or w0, args.height or h0, args.depth or d0
    scene = sc.make(args.scene, width=W, height=H)
    folder = ""
    if scene.get("textures"):
        import tempfile
        folder = tempfile.mkdtemp(prefix="crt_bench_")
        sc.write_bitmaps(scene, folder)
    hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder)
    tracer = pkg.Tracer(hs, device=local_rank, tuning=pkg.tuning_from_string(args.tuning))

    n_tiles = tracer.packed_tile_count(0, 1)
    per_rank = (n_tiles + world - 1) // world            # tiles per rank, padded to the same count
    part_floats = per_rank * 64 * 3
    packed = torch.zeros(part_floats, dtype=torch.float32, device=dev)
    gathered = torch.zeros(part_floats * world, dtype=torch.float32, device=dev) if dist_on else packed
    gather_parts = list(gathered.view(world, part_floats).unbind(0)) if dist_on and rank == 0 else None
    # rank 0 assembles the frame; two buffers, so that frame k's copy to the host can run beside frame k + 1's kernels
    frames2 = [torch.zeros(H * W * 3, dtype=torch.float32, device=dev) for _ in range(2)]
    frame = frames2[0]
    hosts2 = [torch.empty(H * W * 3, dtype=torch.float32).pin_memory() for _ in range(2)] if rank == 0 else None
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    copy_stream = torch.cuda.Stream(dev)
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_copied = [torch.cuda.Event() for _ in range(2)]
    opts = pkg.make_options(depth)

    failed = False
    step_no = [0]

    def step(to_host=False):
        k = step_no[0] % 2
        step_no[0] += 1
        tracer.render_tiles_device(opts, rank, world, packed.data_ptr(), sptr)
        if dist_on and args.rehearse_gloo:
            parts = [torch.empty(part_floats) for _ in range(world)]
            dist.all_gather(parts, packed.cpu())
            gathered.copy_(torch.cat(parts))
        elif dist_on:
            # only rank 0 assembles the frame: a gather to it (point-to-point sends over xGMI), not an all-gather that would deliver the
            # whole frame to every rank
            dist.gather(packed, gather_parts if rank == 0 else None, dst=0)
        if rank == 0:
            if to_host:
                stream.wait_event(ev_copied[k])       # (the copy of the frame that used this buffer two steps ago)
            tracer.unpack_tiles_device(gathered.data_ptr(), world, part_floats, frames2[k].data_ptr(), sptr)
            if to_host:
                # the reference's window ends with the frame in host memory (RayTracer.cpp:289-293): copied on a stream of its own
                ev_done[k].record(stream)
                copy_stream.wait_event(ev_done[k])
                with torch.cuda.stream(copy_stream):
                    hosts2[k].copy_(frames2[k], non_blocking=True)
                    ev_copied[k].record(copy_stream)

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Outside the timed region: (1) one launch of the COUNTING build -- every ray walked the reference's way -- which
    # prices the reference's work for this rank's tiles and whose pixels are the yardstick for (3); (2) one launch of
    # the production kernels tallying the box / triangle tests they actually execute (the roofline's bytes);
    # (3) the production frame, which must equal the counting build's bit for bit or the run fails.
    copts = pkg.make_options(depth, counters=True)
    tracer.render_tiles_device(copts, rank, world, packed.data_ptr(), sptr)
    torch.cuda.synchronize(dev)
    counted_pixels = packed.clone()
    counters = tracer.stats().counters()
    pk_counters, ln_counters = tracer.kernel_counters()
    my_pixels = int(tracer.stats().pixels)
    tracer.render_tiles_device(pkg.make_options(depth, counters=2), rank, world, packed.data_ptr(), sptr)
    torch.cuda.synchronize(dev)
    executed = tracer.executed_counters()
    tracer.render_tiles_device(opts, rank, world, packed.data_ptr(), sptr)
    torch.cuda.synchronize(dev)
    frame_ok = bool(torch.equal(packed.view(torch.int32), counted_pixels.view(torch.int32)))
    del counted_pixels

    def timed(n, to_host):
        fence()
        t_ = time.perf_counter()
        for _ in range(n):
            step(to_host)
        fence()
        e = time.perf_counter() - t_
        if world > 1:
            tt = torch.tensor([e], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        return e

    cdev = torch.device("cpu") if args.rehearse_gloo else dev
    for _ in range(args.warmup):
        step(True)
    fence()
    # queue capacities follow the frames: a context's FIRST frames may outgrow them and be redone by the
    # queue-less kernel.  Such a frame inside the timed region would not be a measurement of the path; before it, it is start-up.
    tracer.synchronize()
    fallback_before = int(tracer.stats().fallback_frames)
    # THE timed region: K frames, each one ending in pinned host memory on rank 0 (frame k's copy beside frame k + 1's kernels)
    elapsed = timed(args.steps, True)
    kernel_ms = tracer.kernel_times_ms(min(args.steps, 64))
    host_ok, host_frame = None, None
    if rank == 0:
        last = (step_no[0] - 1) % 2
        host_ok = bool(torch.equal(hosts2[last].view(torch.int32), frames2[last].cpu().view(torch.int32)))
        host_frame = hosts2[last].numpy().copy()   # what the timed region delivered: compared with the reference's frame below
    # the same loop with the frame left in HBM (reported beside `value`)
    # (the comparisons above leave the GPU idle for tens of milliseconds and its clocks drop: the same W warm-up frames as before `value`'s loop)
    for _ in range(args.warmup):
        step(False)
    resident_elapsed = timed(args.steps, False)

    # The same K steps once more with several frames in flight (reported beside `value`, never as it): frame k on context k % F
    # and stream k % F, each context with the scene resident and queues of its own, so that one frame's latency-bound head and
    # tail (level 0's longest walks; the deeper levels' shadow rays, resolve) overlap its neighbours' busy middle -- what the
    # animation driver does (crt_animation --in-flight).  Every frame is complete inside the timed region and is compared with
    # the one-at-a-time frame afterwards.
    if rank == 0:   # `frame` = the last timed step's tiles (whichever of the two buffers that step used)
        tracer.unpack_tiles_device(gathered.data_ptr(), world, part_floats, frame.data_ptr(), sptr)
        torch.cuda.synchronize(dev)
    pipelined = None
    F = max(0, args.in_flight if args.in_flight is not None else 3)
    if F >= 2:
        try:   # (never lose the line -- `value` -- over the extra measurement: on several GPUs this path is rehearsed, not yet run)
            trs = [tracer] + [pkg.Tracer(hs, device=local_rank, tuning=pkg.tuning_from_string(args.tuning)) for _ in range(F - 1)]
            strs = [torch.cuda.Stream(dev) for _ in range(F)]
            packs = [packed] + [torch.zeros_like(packed) for _ in range(F - 1)]
            gaths = [gathered] + [(torch.zeros_like(gathered) if dist_on else packs[i + 1]) for i in range(F - 1)]
            frames_f = [frame] + [torch.zeros_like(frame) for _ in range(F - 1)] if rank == 0 else [frame] * F
            reference_frame = frame.clone() if rank == 0 else None

            def pstep(k):
                i = k % F
                with torch.cuda.stream(strs[i]):
                    trs[i].render_tiles_device(opts, rank, world, packs[i].data_ptr(), strs[i].cuda_stream)
                    if dist_on and args.rehearse_gloo:
                        strs[i].synchronize()
                        parts = [torch.empty(part_floats) for _ in range(world)]
                        dist.all_gather(parts, packs[i].cpu())
                        gaths[i].copy_(torch.cat(parts))
                    elif dist_on:
                        dist.gather(packs[i], list(gaths[i].view(world, part_floats).unbind(0)) if rank == 0 else None, dst=0)
                    if rank == 0:
                        trs[i].unpack_tiles_device(gaths[i].data_ptr(), world, part_floats, frames_f[i].data_ptr(), strs[i].cuda_stream)

            for k in range(max(args.warmup, 2 * F)):
                pstep(k)
            fence()
            for t_ in trs:
                t_.synchronize()
            fb0 = sum(int(t_.stats().fallback_frames) for t_ in trs)
            tp = time.perf_counter()
            for k in range(args.steps):
                pstep(k)
            fence()
            p_elapsed = time.perf_counter() - tp
            if world > 1:
                t = torch.tensor([p_elapsed], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                p_elapsed = float(t.item())
            for t_ in trs:
                t_.synchronize()
            p_fallback = sum(int(t_.stats().fallback_frames) for t_ in trs) - fb0
            p_ok = None
            if rank == 0:
                p_ok = all(bool(torch.equal(f.view(torch.int32), reference_frame.view(torch.int32))) for f in frames_f)
            pipelined = {"frames_in_flight": F, "value": round(W * H * args.steps / p_elapsed / 1e6, 3), "unit": "Mpixels/s",
                         "ms_per_step": round(p_elapsed / args.steps * 1e3, 4), "frames_match_one_at_a_time": p_ok,
                         "fallback_frames": p_fallback,
                         "note": "the same K steps with frame k on context / stream k % F; every frame complete inside the timed region"}
            del trs[1:], packs[1:], gaths[1:]
        except Exception as e:
            pipelined = {"frames_in_flight": F, "error": repr(e)}

    # every rank's production tiles matched its counting build's?  (and, for N > 1, does the gathered frame equal
    # a single-rank render of the whole frame on rank 0?)
    if world > 1:
        t = torch.tensor([1.0 if frame_ok else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        frame_ok = bool(t.item() > 0.5)
    gathered_ok = None
    if dist_on and rank == 0:
        single = torch.zeros(n_tiles * 192, dtype=torch.float32, device=dev)
        whole = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        tracer.render_tiles_device(opts, 0, 1, single.data_ptr(), sptr)
        tracer.unpack_tiles_device(single.data_ptr(), 1, n_tiles * 192, whole.data_ptr(), sptr)
        torch.cuda.synchronize(dev)
        gathered_ok = bool(torch.equal(whole.view(torch.int32), frame.view(torch.int32)))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * args.steps / elapsed / 1e6
        textured = bool(scene.get("textures"))
        n_k = max(len(kernel_ms), 1)
        avg_kernel_ms = sum(t[0] for t in kernel_ms) / n_k   # first kernel to last
        pk_ms = sum(t[1] for t in kernel_ms) / n_k   # recursion levels (closest-hit walks + shading)
        ln_ms = sum(t[2] for t in kernel_ms) / n_k   # the bulk shadow pass (level-0 shadow rays), HIP events on ITS stream
        rs_ms = sum(t[3] + t[4] for t in kernel_ms) / n_k   # second shadow pass + wave-per-ray tail + resolve
        # SURVEY.md section 8(d) prices a box test at 32 B (24 B box + two 4 B links), a triangle test at 48 B
        # (three positions + the face normal) plus the 4 B leaf index entry that led to it.
        # A box test of a plan loop (kernel_plan.h: the ray against a top-level leaf whose box sits in scalar registers)
        # fetches nothing per ray -- the 64-byte leaf record is loaded once per wave: 1 byte per test.
        def test_bytes(box, tri, plan=0):
            return 32 * box + 52 * tri + plan
        other = algorithmic_bytes(dict(counters, box_tests=0, tri_tests=0, leaf_index_reads=0), 0, 0, textured) + 12 * my_pixels
        b_ref_all = algorithmic_bytes(counters, 0, 0, textured) + 12 * my_pixels   # the REFERENCE's work for this frame
        b_ref_dom = algorithmic_bytes(ln_counters, 0, 0, textured)                  # ... and for the dominant kernel's rays
        b_exec_dom = test_bytes(executed["shadow_pass0_box_tests"], executed["shadow_pass0_tri_tests"], executed["shadow_pass0_plan_tests"])
        b_exec_all = test_bytes(executed["box_tests"], executed["tri_tests"], executed["plan_tests"]) + other
        # The roofline object is for the DOMINANT KERNEL: the bulk shadow pass (kernel_plan.h: stream_trace_shadow_plan<0>
        # on the default path), one launch per frame that walks the level-0 shadow rays.  achieved = the bytes of the tests it EXECUTES (tallied by the kernel itself in the
        # collect_counters == 2 launch above) over its own duration from HIP events on the stream it runs on.
        dom, dom_ms = tracer.kernels().get("shadow0", "stream_trace_shadow"), ln_ms
        achieved = b_exec_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        key = "%s_%dx%d_d%d_n%d" % (args.scene, W, H, depth, world)
        pfile, traffic, evidence_why = committed_evidence(os.path.join(ROOT, "profiles"), key, pkg.csrc_sha256())
        # The bulk shadow pass is launched on a lowest-priority stream beside the recursion levels, whose waves run at
        # s_setprio 3: nothing waits for it until the levels are done, so its duration in the frame is
        # what the levels leave it, not how fast it is.  Its duration ALONE is measured here as well, live, outside the timed
        # region: the same frames on a second context that launches the pass on the caller's stream after the levels
        # (crt_tuning::side_blocks = 0).  `frac` stays the in-frame figure; `alone` is reported beside it.
        alone = None
        try:
            if args.no_alone:
                raise RuntimeError("--no-alone")
            base = dict(kv.split("=") for kv in args.tuning.split()) if args.tuning else {}
            solo = pkg.Tracer(hs, device=local_rank, tuning=pkg.make_tuning(**dict({k: int(v, 0) for k, v in base.items()}, side_blocks=0)))
            for _ in range(4):   # (a context's first frames may outgrow its queues and be redone: time frames after the last such one)
                seen = int(solo.stats().fallback_frames)
                for _ in range(6):
                    solo.render_tiles_device(opts, rank, world, packed.data_ptr(), sptr)
                    solo.synchronize()   # (the capacities follow the PREVIOUS frame's counters, read once that frame is complete)
                if int(solo.stats().fallback_frames) == seen:
                    break
            solo_ms = [t[2] for t in solo.kernel_times_ms(5)]
            del solo
            a_ms = sum(solo_ms) / max(len(solo_ms), 1)
            if a_ms > 0:
                a_gbs = b_exec_dom / (a_ms * 1e-3) / 1e9
                alone = {"kernel_ms": round(a_ms, 4), "achieved": round(a_gbs, 2), "frac": round(a_gbs / HBM_PEAK_GBS, 4),
                         "note": "the same launch with the chip to itself (crt_tuning side_blocks=0: after the recursion levels, on their stream)"}
        except Exception as e:
            alone = {"kernel_ms": None, "note": "not measured: %r" % (e,)}
        # The roofs nearer to these kernels than HBM, from the counters of the committed PMC passes (profiles/r04_pmc.json: per-launch
        # COUNTS, which do not depend on timing) over the launch durations measured LIVE above.  The file records the SHA-256 of the
        # kernel sources it was collected from (tools/collect_profiles.py); when that is not the running tree's, its counts describe
        # other kernels and nothing of it is used: `binding` and `traffic` are null and say why.
        binding, bound, bound_frac, evidence = None, "latency", None, evidence_why
        try:
            if pfile is None:
                traffic = None
            else:
                key = "%s_%dx%d_d%d_n%d" % (args.scene, W, H, depth, world)

                def roofs(pj, ms):
                    """fractions of the nearer roofs for one kernel: counts per launch over the live duration `ms`"""
                    c, clk, t = pj["per_launch"], float(pj["clock_ghz"]) * 1e9, ms * 1e-3
                    t_pmc = float(pj["kernel_ms_in_the_clock_pass"]) * 1e-3   # rocprofv3 serialises launches while it counts
                    r = {"kernel": pj["kernel"], "kernel_ms": round(ms, 4),
                         # a wave64 vector instruction issues 32 lanes per cycle: 2 cycles on one of 1024 SIMDs
                         "valu": round(c["SQ_INSTS_VALU"] * 2.0 / (N_SIMDS * clk * t), 4),
                         # one scalar instruction per cycle and SIMD (the CU's scalar unit serves its four SIMDs in turn: per CU the roof is 4x lower)
                         "salu": round(c["SQ_INSTS_SALU"] / (N_SIMDS * clk * t), 4),
                         "salu_if_one_per_cu_clock": round(c["SQ_INSTS_SALU"] / (N_CUS * clk * t), 4),
                         "l1": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] * 64.0 / t / (N_CUS * 64.0 * clk), 4),
                         "lanes": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0, 4),
                         "in_the_counting_pass": {"kernel_ms": round(t_pmc * 1e3, 4), "waiting": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                                                  "ta_busy": round(c["TA_TA_BUSY_sum"] / (N_CUS * clk * t_pmc), 4),
                                                  "l1_pending_stall": round(c["TCP_PENDING_STALL_CYCLES_sum"] / (N_CUS * clk * t_pmc), 4)}}
                    return r
                binding = {"source": "profiles/r04_pmc.json", "csrc_sha256": str(pfile.get("csrc_sha256"))[:16]}
                if pfile.get(key):
                    binding["dominant"] = roofs(pfile[key], dom_ms)
                    d = binding["dominant"]
                    cands = {"valu issue": d["valu"], "salu issue (one per CU clock)": d["salu_if_one_per_cu_clock"], "vector L1 line rate": d["l1"],
                             "texture-address unit busy": d["in_the_counting_pass"]["ta_busy"],
                             "vector L1 stalled on its miss queue": d["in_the_counting_pass"]["l1_pending_stall"],
                             "latency: waves waiting": d["in_the_counting_pass"]["waiting"]}
                    bound, bound_frac = max(cands.items(), key=lambda kv: kv[1])
                if pfile.get(key + "|levels") and pk_ms > 0:
                    binding["levels"] = roofs(pfile[key + "|levels"], float(pfile[key + "|levels"]["kernel_ms_in_the_clock_pass"]))
                binding["note"] = ("per-launch COUNTS from the committed passes over durations measured live here (levels: over the counted launch's own "
                                   "duration -- in the frame it overlaps the dominant kernel).  valu: 2 cycles per wave64 instruction over 1024 SIMDs; salu: one "
                                   "per cycle and SIMD (and, were it one per CU clock, 4x that); l1: vector-L1 line accesses over 256 CUs x 1 line/clk; lanes: "
                                   "active lanes per vector instruction / 64; in_the_counting_pass: ratios of cycle counters, valid only for the launch they "
                                   "were counted in")
        except Exception as e:
            evidence = "profiles/r04_pmc.json not usable: %r" % (e,)
            traffic = None
        out = {
            "metric": "Mpixels/s at 1920x1080 depth 8; HBM GB/s vs roofline",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d triangles, %dx%d, depth %d, 8x8 tiles dealt round-robin over %d GPU(s)"
                                   % (sc.DESCRIPTIONS.get(args.scene, args.scene), sc.triangle_count(scene), W, H, depth, world),
                       "scene": args.scene, "width": W, "height": H, "max_depth": depth,
                       "parallelism": "tiles8x8-roundrobin-%d" % world, "tuning": args.tuning or "defaults"},
            "frame_matches_counting_build": frame_ok,
            "pipelined": pipelined,
            "value_device_resident": round(W * H * args.steps / resident_elapsed / 1e6, 3),   # the same K steps without the copy to the host
            "host_frame_matches_device": host_ok,
            "kernel_ms": {"first_to_last": round(avg_kernel_ms, 4), "recursion_levels": round(pk_ms, 4),
                          "shadow_pass0_overlapped": round(ln_ms, 4), "shadow_pass1_heavy_resolve": round(rs_ms, 4)},
            "roofline": {"bound": bound, "bound_frac": bound_frac, "evidence": evidence, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "peak_measured": HBM_MEASURED_GBS, "frac_of_measured": round(achieved / HBM_MEASURED_GBS, 4),
                         "kernel": dom, "kernel_ms": round(dom_ms, 4), "alone": alone, "binding": binding,
                         "executed_bytes_per_launch": int(b_exec_dom),
                         "executed": {"box_tests": executed["shadow_pass0_box_tests"], "tri_tests": executed["shadow_pass0_tri_tests"],
                                      "plan_tests": executed["shadow_pass0_plan_tests"]},
                         "bound_note": "`bound` names the unit with the highest fraction among the roofs of `binding` (vector issue, scalar issue, "
                                       "vector-L1 line rate, texture-address busy, L1 miss-queue stall, waves waiting) -- 'latency' when no fresh counters "
                                       "are at hand.  `achieved` / `frac` stay what SURVEY.md section 8(d) prescribes for this path, priced against HBM: "
                                       "the bytes of the box / triangle tests the kernel EXECUTES (32 B per child box of a filter node, 52 B per "
                                       "triangle) over its duration IN THE FRAME, where it runs at the lowest priority beside the recursion levels "
                                       "(`alone`: with the chip to itself).  HBM does not bound it: the scene lives in L2 / Infinity Cache and `traffic` "
                                       "(PMC, fabric side) is a fraction of the executed bytes -- profiles/, DESIGN.md section 4",
                         "reference_work": {
                             "note": "what the REFERENCE does for the same rays (counting build == the oracle's counters).  The production "
                                     "kernel skips work that cannot change the result (one walk per mesh and ray, exact shadow early exit), "
                                     "so this rate is NOT a bandwidth and may exceed the HBM peak",
                             "bytes_per_launch": int(b_ref_dom), "counters": ln_counters,
                             "reference_work_rate_gbs": round(b_ref_dom / (dom_ms * 1e-3) / 1e9, 2) if dom_ms > 0 else None},
                         "whole_frame": {"executed_bytes": int(b_exec_all), "kernels_ms": round(avg_kernel_ms, 4),
                                         "achieved": round(b_exec_all / (avg_kernel_ms * 1e-3) / 1e9, 2),
                                         "frac": round(b_exec_all / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "executed": {"box_tests": executed["box_tests"], "tri_tests": executed["tri_tests"],
                                                      "plan_tests": executed["plan_tests"]},
                                         "reference_work_bytes": int(b_ref_all),
                                         "reference_work_rate_gbs": round(b_ref_all / (avg_kernel_ms * 1e-3) / 1e9, 2),
                                         "reference_counters": counters}},
        }
        if gathered_ok is not None:
            out["gathered_frame_matches_single_rank"] = gathered_ok
        out["collective"] = ("gloo (rehearsal)" if args.rehearse_gloo else "rccl gather to rank 0") if dist_on else None
        tracer.synchronize()
        fallback = int(tracer.stats().fallback_frames) - fallback_before
        out["fallback_frames"] = fallback                       # in the timed region (and the copy-inclusive one after it)
        out["fallback_frames_before_timing"] = fallback_before  # start-up frames redone while the queue capacities settled
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sc, args.scene, depth, W, H, gpu_frame=host_frame)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
        cb = out.get("cpu_baseline") or {}
        if cb.get("frame_matches_gpu") is False:
            print("bench.py: FAILED self-check: the GPU frame differs from the reference's own frame in %d floats"
                  % cb.get("differing_floats", -1), file=sys.stderr, flush=True)
            failed = True
        if pipelined and pipelined.get("frames_match_one_at_a_time") is False:
            print("bench.py: FAILED self-check: a frame of the pipelined run differs from the one-at-a-time frame", file=sys.stderr, flush=True)
            failed = True
        if not frame_ok or gathered_ok is False or fallback:
            # a line whose frame is wrong (or that silently took the 4x slower fallback) is not a measurement
            print("bench.py: FAILED self-check: frame_matches_counting_build=%r gathered_frame_matches_single_rank=%r "
                  "fallback_frames=%d" % (frame_ok, gathered_ok, fallback), file=sys.stderr, flush=True)
            failed = True
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
