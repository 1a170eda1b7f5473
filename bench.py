#!/usr/bin/env python3
"""bench.py — throughput of the H.264 hot path on MI355X (BASELINE.json config 2:
"H.264 8-bit 1080p: idct_add + qpel MC + deblock on 1xMI355X").

A step = one pass of the hot path (reconstruct + deblock) over a batch of `--frames`
independent 1080p pictures (= independent streams) per GPU, inputs already
resident in HBM.  One process per GPU; ranks share nothing on the data path (weak scaling,
streams shard across ranks); torch.distributed (RCCL) is only the control plane: barrier and
max/sum of the per-rank counters.

Prints ONE JSON line: metric macroblocks/s (whole job), plus `roofline` for the dominant
kernel (HIP-event timed), `cpu_baseline` (the reference's own C functions from oracle/_ref/libref.so —
kind "reference" — or, where that object is missing, the scalar oracle — kind "port" — on ALL hardware threads of
this box, pinned, on a bounded sample of the same workload) and `extra`: further measured points of the same path
(64 pictures per step, all-intra pictures, content on which the loop filter fires, swscale config 5, HEVC config 3).

The workload generator is the one tests/test_frame_gpu.py::test_full_size_1080p_batch_matches_oracle checks
bit-exactly against the oracle at full size; the bench itself does not check outputs.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic bytes per macroblock: each input byte read once, each output written once
B_RECON = 384 + 768 + 128 + 384          # reference samples + coefficients + MB record/mv + unfiltered write
B_DEBLOCK = 384 + 384 + 64               # the loop filter alone: unfiltered read + filtered write + side info (its isolated roofline figure)
B_FUSED_HIGH10 = 4736                    # the same accounting with 16-bit samples (4 x 384 more) and 32-bit coefficients (768 more)
B_FUSED = 2432                           # SURVEY.md 8(d) / DESIGN 6.1: the two-surface pipeline figure the headline fraction is quoted on
                                         # (1664 + 768: the filter's second read of the 64-byte record is not part of the contract figure)
HBM_PEAK = 8.0e12
LAYOUT_MASK = {False: 1, True: 2}        # MI355_LAYOUTS_LINEAR / MI355_LAYOUTS_TILED: the bench knows which layout its pictures have


def level_widths(fs):
    """host array for mi355_h264_recon_intra_all_dev / _levels_dev: widest level l over the pictures of the batch"""
    import ctypes as C
    return (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])


def measured_traffic(kernel, frames):
    """HBM bytes per launch of `kernel` from the PMC passes of the session named in profiles/CURRENT_TRAFFIC.json (a tracked file: {"file": ..., "session": ...});
    the passes cannot be collected from inside this process (tools/gpu_traffic.sh, tools/mk_traffic_profile.py)"""
    try:
        cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT_TRAFFIC.json")))
        path = os.path.join(ROOT, "profiles", cur["file"])
        t = json.load(open(path))
        if abs(t.get("frames_per_gpu", -9) - frames) <= 1 and kernel in t.get("kernels", {}):      # 2048 pictures as three launches: 683 / 683 / 682
            return t["kernels"][kernel]["traffic_bytes"], os.path.relpath(path, ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


# The driver's record keeps the scalars of `config` under keys cut at 40 characters, about twenty of them: the points it should carry, in this order,
# under these (<= 32 characters, unique) names.  Everything else stays in the line's `extra` (and in profiles/ as the session's full line).
POINT_KEYS = [
    ("config2_f%(F)d", "p_c2_f%(F)d"), ("config2_f2048_one_pipeline", "p_c2_one_pipeline"), ("config3_hevc_2160p10_chain", "p_c3_hevc_chain"),
    ("config3_hevc_2160p10_chain:one_chain", "p_c3_hevc_one_chain"), ("config5_sws_hd_special", "p_c5_sws_special"), ("config5_sws_hd_generic", "p_c5_sws_generic"),
    ("config5_sws_uhd_to_hd", "p_c5_sws_uhd_to_hd"), ("config2_f64", "p_c2_f64"), ("config2_f512", "p_c2_f512"), ("config2_smooth_f2048", "p_c2_smooth"),
    ("config2_mixed_partitions_f2048", "p_c2_mixed_parts"), ("config2_high10_f2048", "p_c2_high10"), ("all_intra_f512", "p_all_intra_f512"),
    ("config2_f2048_detile", "p_c2_detile"), ("hevc_bridge_pb_1080p_few_intra_bridge_x16", "p_hevc_br_x16"), ("hevc_bridge_pb_1080p_few_intra_c_x16", "p_hevc_c_x16"),
    ("h264_bridge_1080p_x64", "p_h264_br_x64"), ("h264_bridge_1080p_x64_c", "p_h264_c_x64"), ("hevc_bridge_i_ctb64:forced", "p_hevc_i_on_device"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=2048, help="independent 1080p pictures (streams) per GPU per step")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic pictures generated on the host; "
                    "they are replicated (own copies in HBM) to fill --frames")
    ap.add_argument("--pipelines", type=int, default=3, help="the step's batch as this many independent pipelines (shares of the pictures as equal as they come), each with its own "
                    "HIP stream through the three passes: one pipeline's loop filter (bound by the vector pipe) runs beside another's reconstruction (bound by the memory "
                    "pipeline), and the reconstruction launches take turns (--no-phased: they do not).  3 is the default since round 5's last sessions (profiles/r05y_pipelines.txt, one box, "
                    "alternating: 12.0 - 12.2 ms per step against 12.7 for 2 and 12.9 - 13.7 for 1; 4 and more are slower again); 1 = the three passes over the whole batch one after the "
                    "other (how rounds 1-4 measured; the extra point config2_f2048_one_pipeline keeps measuring that)")
    ap.add_argument("--no-phased", dest="phased", action="store_false", help="with --pipelines > 1: let the pipelines' reconstruction launches start whenever their streams get to them "
                    "(default: they take turns — pipeline p's waits for pipeline p - 1's, the first one's for the last one's of the step before — so a reconstruction never runs beside "
                    "another reconstruction, only beside the other pipelines' loop filters)")
    ap.add_argument("--mb-width", type=int, default=120)
    ap.add_argument("--mb-height", type=int, default=68)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-extra", action="store_true", help="skip the additional measured points (rank 0, N=1 only)")
    ap.add_argument("--no-alone", action="store_true", help="skip the three whole-batch launches of the dominant kernel after the timed region (roofline.alone_*): what the "
                    "counter passes use (tools/gpu_traffic.sh), so that every launch of the profiled command is a share-sized one")
    ap.add_argument("--notes", action="store_true", help="keep the long `note` / `sample` texts of the extra points in the JSON line (without them "
                    "the whole line stays under the 16 KB the driver's record keeps; what each point is: README.md, DESIGN.md 6)")
    ap.add_argument("--layout", choices=("tiled", "linear"), default="tiled", help="surface layout of dst / recon / reference "
                    "pictures in HBM: macroblock-tiled (the decoded-picture-buffer layout, include/mi355_h264_frame.h) or planes "
                    "with line strides (AVFrame-like)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend of the control plane at N>1 "
                    "(nccl = RCCL; gloo only with --dry, for the CPU test of the launch path)")
    ap.add_argument("--dry", action="store_true", help="control plane only: no GPU, no library; a step is a 1 ms sleep.  What the CPU test of "
                    "`--gpus N` uses (tests/test_bench_launch.py); the line says \"dry\": true and is not a measurement")
    ap.add_argument("--queue", action="store_true", help="N>1: ranks pull step-sized batches of streams from the work queue "
                    "(libav_amd.shard.WorkQueue) instead of the static deal; a rank may then run more or fewer than --steps steps")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N ranks and not started by a launcher: start them (one process per GPU, the contract's own command line)
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the two must agree" % (args.gpus, world))
    if args.backend == "gloo" and not args.dry:
        sys.exit("bench.py: --backend gloo is the CPU control-plane test and needs --dry")
    ctl = "cpu" if (world == 1 or args.backend == "gloo") else "cuda"      # where the control plane's small tensors live
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if args.dry:
        return dry_run(args, rank, world, dist, ctl)

    import numpy as np
    import libav_amd
    import h264_frames as HF
    lib = libav_amd.load(local_rank)          # raises if the HIP library / GPU is missing: no fallback

    class Prov:                                # the tiny provider interface h264_frames.DeviceFrames expects
        pass
    prov = Prov()
    prov.lib = lib

    mbw, mbh, F = args.mb_width, args.mb_height, args.frames
    nmb = mbw * mbh
    G = max(1, min(args.distinct, F))
    # control plane: rank 0 owns the stream table (world*F streams), every rank takes s % world == rank
    from libav_amd import shard
    n_streams = world * F
    table = shard.make_stream_table(n_streams, 0x264) if rank == 0 else None
    table = shard.broadcast_stream_table(table, n_streams, ctl)
    mine = shard.my_streams(table, rank, world)
    assert len(mine) == F
    fs = HF.synth_frames_fast(G, mbw, mbh, seed=mine[0][1], lib=lib)
    # the G distinct pictures are replicated ON THE DEVICE to F pictures, each with its own buffers in HBM
    tiled = args.layout == "tiled"
    dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=tiled)
    big = fs

    for name, res, at in (("mi355_h264_recon_inter_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_h264_recon_intra_all_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                          ("mi355_h264_deblock_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_h264_decode_frames_wide_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_event_create", C.c_void_p, []), ("mi355_event_record", C.c_int, [C.c_void_p, C.c_void_p]),
                          ("mi355_event_elapsed_ms", C.c_float, [C.c_void_p, C.c_void_p]), ("mi355_sync", C.c_int, [C.c_void_p])):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = at
    # the batch as P pipelines of about F / P pictures, each on its own stream (P = 1: the null stream)
    P = args.pipelines if args.pipelines > 0 and F // args.pipelines >= 1 else 1
    lib.mi355_stream_create.restype = C.c_void_p
    streams = [None]
    counts = [F // P + (1 if i < F % P else 0) for i in range(P)]         # 2048 as 683 + 683 + 682
    firsts = [sum(counts[:i]) for i in range(P)]
    per = counts[0]                                                       # the (largest) launch the line's per-launch figures are quoted on
    frame_bytes = C.sizeof(dev.host_desc) // F

    # Several pipelines: the library's own object (libav_amd/csrc/h264_pipelines.hip: a stream per share, the shares' reconstruction launches taking turns, events around
    # every pass) — the schedule is product code, this file only calls it.  One pipeline (and the work-queue mode): the three entry points on the null stream.
    pipe = None
    if P > 1 and not (args.queue and world > 1):
        lib.mi355_h264_pipelines_create.restype = C.c_void_p
        lib.mi355_h264_pipelines_create.argtypes = [C.c_int, C.c_int]
        lib.mi355_h264_pipelines_decode_dev.restype = C.c_int
        lib.mi355_h264_pipelines_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.mi355_h264_pipelines_sync.argtypes = [C.c_void_p]
        lib.mi355_h264_pipelines_timing.argtypes = [C.c_void_p, C.c_int]
        lib.mi355_h264_pipelines_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mi355_h264_pipelines_destroy.argtypes = [C.c_void_p]
        pipe = C.c_void_p(lib.mi355_h264_pipelines_create(P, 1 if args.phased else 0))
        assert pipe
        lib.mi355_h264_pipelines_timing(pipe, 1)
    elif P > 1:
        P, counts, firsts, per = 1, [F], [0], F

    def step(events=None):
        if pipe is not None:
            assert lib.mi355_h264_pipelines_decode_dev(pipe, C.c_void_p(dev.d_desc), F, mbw, mbh, big.max_intra_level, level_widths(big), LAYOUT_MASK[tiled]) == 0
            return
        for p_, st in enumerate(streams):
            d = C.c_void_p(dev.d_desc + firsts[p_] * frame_bytes)
            n_ = counts[p_]
            ev = events[p_] if events is not None else None
            if ev is not None:
                lib.mi355_event_record(ev[0], st)
            assert lib.mi355_h264_recon_inter_layouts_dev(d, n_, mbw, mbh, LAYOUT_MASK[tiled], st) == 0
            if ev is not None:
                lib.mi355_event_record(ev[1], st)
            assert lib.mi355_h264_recon_intra_all_dev(d, n_, mbw, mbh, big.max_intra_level, level_widths(big), st) == 0
            if ev is not None:
                lib.mi355_event_record(ev[2], st)
            assert lib.mi355_h264_deblock_layouts_dev(d, n_, mbw, mbh, LAYOUT_MASK[tiled], st) == 0
            if ev is not None:
                lib.mi355_event_record(ev[3], st)

    def sync_all():
        if pipe is not None:
            assert lib.mi355_h264_pipelines_sync(pipe) == 0
        for st in streams:
            assert lib.mi355_sync(st) == 0

    def barrier():
        if dist is not None:
            dist.barrier()
        sync_all()

    for _ in range(args.warmup):
        step()
    if pipe is not None:
        assert lib.mi355_h264_pipelines_collect(pipe, None, None) == 0         # the warm-up's events
    queue = shard.WorkQueue(world * args.steps, 1) if (args.queue and world > 1) else None
    evs = []
    barrier()
    t0 = time.perf_counter()
    def new_events():
        return [[lib.mi355_event_create() for _ in range(4)] for _ in range(P)]
    if queue is None:
        for k in range(args.steps):
            evs.append(new_events() if pipe is None else None)
            step(evs[-1])
    else:
        # one batch in flight plus one queued: a rank only pulls when its previous-but-one batch has finished
        while True:
            if len(evs) >= 2:
                for e in evs[-2]:
                    lib.mi355_event_elapsed_ms(e[0], e[3])      # waits for that batch's last events
            if queue.next() is None:
                break
            evs.append(new_events())
            step(evs[-1])
    sync_all()
    barrier()
    elapsed = time.perf_counter() - t0
    my_steps = len(evs)

    # per LAUNCH (one pipeline's share of the batch): with P > 1 the launches of different pipelines run side by side, so the passes' times do not add up to the step
    if pipe is not None:
        sums, nl_ = (C.c_double * 3)(), C.c_int(0)
        assert lib.mi355_h264_pipelines_collect(pipe, sums, C.byref(nl_)) == 0
        assert nl_.value == my_steps * P, (nl_.value, my_steps, P)
        t_inter, t_intra, t_deblock = (sums[k] / max(1, nl_.value) for k in range(3))
    else:
        nl = max(1, my_steps * P)
        t_inter = sum(lib.mi355_event_elapsed_ms(e[0], e[1]) for st_ in evs for e in st_) / nl
        t_intra = sum(lib.mi355_event_elapsed_ms(e[1], e[2]) for st_ in evs for e in st_) / nl
        t_deblock = sum(lib.mi355_event_elapsed_ms(e[2], e[3]) for st_ in evs for e in st_) / nl

    elapsed, total_mbs = shard.reduce_counters(elapsed, F * nmb * my_steps, ctl)
    steps_per_rank = shard.gather_counts(my_steps, ctl)
    backend_world = dist.get_world_size() if dist is not None else 1

    if rank == 0:
        value = total_mbs / elapsed
        n_intra = int(sum(len(big.intra_list[f % G]) for f in range(F)))
        n_inter = F * nmb - n_intra
        # dominant kernel = the pass with the largest share of the step (the loop filter is ONE launch for all bands of all pictures since round 4)
        passes = {"k_recon_inter_tiled" if tiled else "k_recon_inter": (t_inter, P, n_inter * B_RECON),
                  "k_deblock_tiled" if tiled else "k_deblock_linear": (t_deblock, P, F * nmb * B_DEBLOCK)}
        # dominant kernel = the one that moves the most algorithmic bytes per step (with one pipeline also the longest pass; with several, launches share the device and
        # a launch's duration says how long it was resident, not how much of the device it had)
        dom = max(passes, key=lambda k: passes[k][2])
        t_pass, launches, bytes_pass = passes[dom]               # t_pass: the average duration of ONE launch (HIP events on its stream)
        achieved = bytes_pass / launches / (t_pass * 1e-3)       # algorithmic bytes per launch / avg launch time
        out = {
            "metric": "macroblocks_per_s", "value": value, "unit": "macroblocks/s",
            "frames_per_s": value / nmb, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "rccl_world_size": backend_world, "distribution": "work queue" if queue is not None else "static deal",
            "steps_per_rank": steps_per_rank,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "H.264 8-bit 4:2:0 1080p (1920x1088 coded), P pictures: qpel MC + idct_add + deblock, "
                                   "two-surface pipeline, %d independent pictures per GPU per step as %d pipeline(s) of %s pictures on their own HIP streams%s "
                                   "(%d distinct synthetic pictures replicated), 5%% Intra16x16 MBs" % (F, P, " / ".join(str(c) for c in counts), ", reconstruction launches taking turns" if (args.phased and P > 1) else "", G),
                       "surface_layout": "macroblock-tiled decoded-picture-buffer surfaces (256-byte luma + 128-byte chroma tiles; "
                                         "a picture is de-tiled only when it leaves HBM: extra point config2_f2048_detile)" if tiled
                                         else "planes with line strides",
                       "frames_per_gpu": F, "pipelines": P, "phased": bool(args.phased and P > 1), "frames_per_launch": per, "mb_per_frame": nmb, "bytes_per_mb_fused": B_FUSED,
                       "fused_fraction_of_hbm_roofline": value / world * B_FUSED / HBM_PEAK,
                       "parallelism": "independent streams sharded over %d GPU(s), no data-path collective" % world,
                       "verified_by": "tests/test_frame_gpu.py::test_full_size_1080p_batch_run_kernel + ::test_pipelines_object_at_the_bench_share (same generator, tiled surfaces, the run kernel, the pipelines object at a 683-picture share; every sample)"},
            "pass_ms": {"recon_inter": t_inter, "recon_intra": t_intra, "deblock": t_deblock},
            "pass_ms_is": "average duration of one launch (%d pictures) by HIP events on its stream%s" % (per, "; the %d pipelines' launches overlap: the passes do not add up to ms_per_step" % P if P > 1 else ""),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": None,
                         "launches_per_step": launches, "avg_launch_us": t_pass * 1e3,
                         "algorithmic_bytes_per_launch": bytes_pass / launches},
        }
        if P > 1 and not args.no_alone:
            # the same kernel ALONE, outside the timed region: one launch over the whole batch, nothing beside it (what rounds 1-4's lines and the extra point
            # config2_f2048_one_pipeline measure)
            e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
            sync_all()
            alone = []
            for _ in range(3):
                lib.mi355_event_record(e0, None)
                assert lib.mi355_h264_recon_inter_layouts_dev(C.c_void_p(dev.d_desc), F, mbw, mbh, LAYOUT_MASK[tiled], None) == 0
                lib.mi355_event_record(e1, None)
                alone.append(lib.mi355_event_elapsed_ms(e0, e1))
            t_alone = sum(alone[1:]) / 2
            out["roofline"]["alone_us"] = t_alone * 1e3            # the same kernel, one launch over all F pictures with nothing beside it (after the timed region)
            out["roofline"]["alone_frac"] = n_inter * B_RECON / (t_alone * 1e-3) / HBM_PEAK
            out["roofline"]["alone_frames"] = F
            # a launch shares the device with the other pipeline's launches: its duration is the time it was resident, not the time it would take alone
            out["roofline"]["shared_device"] = ("%d pipelines: this kernel's launch (%d pictures) runs beside the other pipelines' loop filters and intra passes, so achieved / frac are per launch WHILE SHARING the device; "
                                               "the same kernel with the device to itself: roofline.alone_frac / alone_us (and extra point config2_f2048_one_pipeline); the whole job's rate is "
                                               "config.fused_fraction_of_hbm_roofline" % (P, per))
        # HBM bytes per launch of the dominant kernel from the PMC passes of the same command
        # (tools/gpu_traffic.sh -> profiles/*hbm_traffic*.json; cannot be collected from inside this process)
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(dom, per)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fs, args.cpu_seconds)
        dev.free()
        dev = None
        # every measured point's fraction of the HBM roofline as a short scalar inside `config` (the driver's record keeps
        # `config` whole and only the NAMES of other top-level keys): headline first
        points = {"config2_f%d" % F: round(value / world * B_FUSED / HBM_PEAK, 4)}
        if world == 1 and not args.no_extra:
            out["extra"] = extra_points(lib, prov, mbw, mbh, tiled)
            for p in out["extra"]:
                for key in ("fused_fraction_of_hbm_roofline", "fraction_of_hbm_roofline"):
                    if isinstance(p.get(key), float):
                        points[p["name"]] = round(p[key], 4)
                for key in ("bridge", "hooked", "reference_c_decoder"):            # decoder end to end: pictures/s, bridge (or hooked tables) vs C
                    if isinstance(p.get(key), dict) and "pictures_per_s" in p[key]:
                        points[p["name"] + ("_c" if key == "reference_c_decoder" else "")] = round(p[key]["pictures_per_s"], 1)
                if isinstance(p.get("one_pipeline"), dict) and "fraction_of_hbm_roofline" in p["one_pipeline"]:      # config 3 as ONE chain (the point itself: two chains on streams)
                    points[p["name"] + ":one_chain"] = round(p["one_pipeline"]["fraction_of_hbm_roofline"], 4)
                if isinstance(p.get("bridge_forced"), dict) and "pictures_per_s" in p["bridge_forced"]:
                    points[p["name"] + ":forced"] = round(p["bridge_forced"]["pictures_per_s"], 1)
                for key in p:                                                       # ... with several decoders in the process
                    if isinstance(p[key], dict) and "pictures_per_s" in p[key] and (key.startswith("bridge_x") or key.startswith("reference_c_decoder_x")):
                        points[p["name"] + "_" + key.replace("reference_c_decoder", "c")] = round(p[key]["pictures_per_s"], 1)
            if not args.notes:
                for p in out["extra"]:
                    for key in ("note", "what", "sample", "pass_ms_is"):
                        p.pop(key, None)
                    if isinstance(p.get("cpu_baseline"), dict):
                        p["cpu_baseline"].pop("sample", None)
        for long_, short_ in POINT_KEYS:            # flat (the driver's record keeps the scalars of `config`, not nested objects), the named ones only, in this order
            long_, short_ = long_ % {"F": F}, short_ % {"F": F}
            if long_ in points:
                assert len(short_) <= 32
                out["config"][short_] = points[long_]
        if args.notes:
            out["config"]["points"] = points
        print(json.dumps(out))
    if dev is not None:
        dev.free()
    if dist is not None:
        dist.destroy_process_group()


def dry_run(args, rank, world, dist, ctl):
    """--dry: the launch path and the control plane of the bench with nothing behind them (CPU test of `--gpus N`)."""
    from libav_amd import shard
    F, nmb = args.frames, args.mb_width * args.mb_height
    table = shard.make_stream_table(world * F, 0x264) if rank == 0 else None
    table = shard.broadcast_stream_table(table, world * F, ctl)
    assert len(shard.my_streams(table, rank, world)) == F
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    if dist is not None:
        dist.barrier()
    elapsed, total = shard.reduce_counters(time.perf_counter() - t0, F * nmb * args.steps, ctl)
    steps_per_rank = shard.gather_counts(args.steps, ctl)
    if rank == 0:
        print(json.dumps({"metric": "macroblocks_per_s", "value": total / elapsed, "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "rccl_world_size": dist.get_world_size() if dist is not None else 1, "steps_per_rank": steps_per_rank,
                          "vs_baseline": None, "dtype": "u8", "data": "none", "dry": True,
                          "config": {"workload": "DRY RUN: control plane only (%s), no kernel ran" % args.backend}}))
    if dist is not None:
        dist.destroy_process_group()


def extra_points(lib, prov, mbw, mbh, tiled=True):
    """Further measured points of the same path (each: the three passes back to back, HIP events, 3 steps after 1 warm-up)."""
    import h264_frames as HF
    pts = []

    def run(name, fs, F, note, layout_tiled=tiled, detile=False, pipelines=1):
        dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=layout_tiled)
        try:
            conv = detile_jobs(lib, dev, fs, F) if detile else None
            # a batch of I pictures: the host knows (slice types) that no picture holds an inter macroblock and does not launch that pass
            # (what the bridge's dispatcher does for such a launch set); the descriptors say the same (MI355_FRAME_NO_INTER)
            no_inter = all(int(fs.intra_start[g][-1]) == fs.mb_w * fs.mb_h for g in range(fs.F))

            def inter():
                if not no_inter:
                    assert lib.mi355_h264_recon_inter_layouts_dev(dev.d_desc, F, mbw, mbh, LAYOUT_MASK[layout_tiled], None) == 0

            def once():
                inter()
                assert lib.mi355_h264_recon_intra_all_dev(dev.d_desc, F, mbw, mbh, fs.max_intra_level, level_widths(fs), None) == 0
                assert lib.mi355_h264_deblock_layouts_dev(dev.d_desc, F, mbw, mbh, LAYOUT_MASK[layout_tiled], None) == 0
                if conv is not None:
                    assert lib.mi355_h264_surface_convert_dev(conv, F, mbw, mbh, None) == 0
            once()
            if pipelines > 1:
                # the headline's execution: the library's pipelines object (shares on their own streams, reconstruction launches taking turns)
                lib.mi355_h264_pipelines_create.restype = C.c_void_p
                lib.mi355_h264_pipelines_create.argtypes = [C.c_int, C.c_int]
                lib.mi355_h264_pipelines_decode_dev.restype = C.c_int
                lib.mi355_h264_pipelines_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
                lib.mi355_h264_pipelines_sync.argtypes = [C.c_void_p]
                lib.mi355_h264_pipelines_destroy.argtypes = [C.c_void_p]
                pipe = C.c_void_p(lib.mi355_h264_pipelines_create(pipelines, 1))
                assert pipe
                lw = level_widths(fs)

                def piped():
                    assert lib.mi355_h264_pipelines_decode_dev(pipe, C.c_void_p(dev.d_desc), F, mbw, mbh, fs.max_intra_level, lw, LAYOUT_MASK[layout_tiled]) == 0
                lib.mi355_sync(None)
                piped(); piped()
                assert lib.mi355_h264_pipelines_sync(pipe) == 0
                t0 = time.perf_counter()
                for _ in range(4):
                    piped()
                assert lib.mi355_h264_pipelines_sync(pipe) == 0
                ms = (time.perf_counter() - t0) * 1e3 / 4
                lib.mi355_h264_pipelines_destroy(pipe)
            else:
                e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
                lib.mi355_event_record(e0, None)
                for _ in range(3):
                    once()
                lib.mi355_event_record(e1, None)
                ms = lib.mi355_event_elapsed_ms(e0, e1) / 3
            # one more step with an event after every pass: where the step's time goes
            ev = [lib.mi355_event_create() for _ in range(4)]
            lib.mi355_event_record(ev[0], None)
            inter()
            lib.mi355_event_record(ev[1], None)
            assert lib.mi355_h264_recon_intra_all_dev(dev.d_desc, F, mbw, mbh, fs.max_intra_level, level_widths(fs), None) == 0
            lib.mi355_event_record(ev[2], None)
            assert lib.mi355_h264_deblock_layouts_dev(dev.d_desc, F, mbw, mbh, LAYOUT_MASK[layout_tiled], None) == 0
            lib.mi355_event_record(ev[3], None)
            lib.mi355_sync(None)
            passes = {k: lib.mi355_event_elapsed_ms(ev[i], ev[i + 1]) for i, k in enumerate(("recon_inter", "recon_intra", "deblock"))}
        finally:
            dev.free()
        v = F * mbw * mbh / (ms * 1e-3)
        pts.append({"name": name, "macroblocks_per_s": v, "frames_per_s": v / (mbw * mbh), "ms_per_step": ms, "frames_per_step": F, "pipelines": pipelines,
                    "fused_fraction_of_hbm_roofline": v * B_FUSED / HBM_PEAK, "pass_ms": passes,
                    "pass_ms_is": "one further step, the three passes one after the other on one stream" + (" (ms_per_step: %d pipelines as in the headline)" % pipelines if pipelines > 1 else ""),
                    "note": note})

    base = HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib)
    for fn in (hevc_point, hevc_bridge_points, sws_points, session_points, h264_real_stream_points):
        try:
            r = fn(lib)
            pts.extend(r if isinstance(r, list) else [r])
        except Exception as e:                 # an extra point must not take the headline line down with it
            pts.append({"name": fn.__name__, "error": repr(e)})
    # the config-2 family LAST on the line (the driver's record keeps the line's last 16 KB)
    if tiled:
        run("config2_f2048_detile", base, 2048, "the headline workload with every finished picture also converted to planes with line strides "
            "(mi355_h264_surface_convert_dev: what a picture costs when it LEAVES HBM — display, host copy, a consumer that wants lines; "
            "+384 B read and written per macroblock; references stay tiled)", detile=True)
    run("config2_f2048_linear" if tiled else "config2_f2048_tiled", base, 2048, "the headline workload on the OTHER surface layout (%s)"
        % ("planes with line strides, as rounds 1-2 measured" if tiled else "macroblock-tiled"), layout_tiled=not tiled)
    mixed = HF.synth_frames_fast(4, mbw, mbh, seed=0x2640, lib=lib, partitions="mixed")
    run("config2_mixed_partitions_f2048", mixed, 2048, "SURVEY 8d's second run of config 2: inter macroblocks are 16x16 / 16x8 / 8x16 / 8x8 (a quarter each), 8x8 "
        "quadrants 8x8 / 8x4 / 4x8 / 4x4 (a quarter each), one vector per partition, one reference per partition / quadrant: 5.6 prediction blocks "
        "and reference windows per macroblock on average instead of 1 (the algorithmic bytes stay 2432 per macroblock: the fraction is against the "
        "same figure); verified by tests/test_frame_gpu.py::test_full_size_1080p_mixed_partitions_matches_oracle")
    run("config2_f2048_one_pipeline", base, 2048, "the headline batch as ONE pipeline on one stream: the three passes over all 2048 pictures one after the other "
        "(how rounds 1-4 measured the headline; pass_ms here are the passes' own times)")
    # small batches run as the headline does — three shares through the library's pipelines object (profiles/r06_experiments.md 6: 512 pictures 35.9 -> 40.3 %,
    # 64 pictures 15.7 -> 17.7 %: there the loop filter's chain of dependent steps per picture is the step)
    run("config2_f64", base, 64, "SURVEY 8d's stated batch: 64 pictures per step, three shares on streams ", pipelines=3)
    run("config2_f512", base, 512, "512 pictures per step, three shares on streams", pipelines=3)
    intra = HF.synth_frames_fast(2, mbw, mbh, seed=0x1264, lib=lib, intra_frac=1.0)
    run("all_intra_f512", intra, 512, "I pictures: every macroblock Intra16x16, %d dependency levels = launches of k_recon_intra; the inter pass is not launched for a batch of I pictures" % intra.max_intra_level)
    smooth = HF.synth_frames_fast(4, mbw, mbh, seed=0x2264, lib=lib, refs="smooth", coef_b=4)
    run("config2_smooth_f2048", smooth, 2048, "same shapes, smooth reference pictures and small residuals: the loop filter's conditions "
        "hold on most lines (on config 2's random references they almost never do and the wave-level early-outs skip the arithmetic)", pipelines=3)
    # High 10: the same workload with 10-bit samples and 32-bit coefficients through the second kernel set — last: its 2048 pictures take
    # 100 GB of HBM, and the decoder processes of the real-stream points above should not start beside an allocator that has just let go of them
    try:
        F10 = 2048
        dev = HF.DeviceFrames(prov, base, replicate=F10, bit_depth=10)
        try:
            lw = level_widths(base)

            def wide(passes):
                assert lib.mi355_h264_decode_frames_wide_dev(dev.d_desc, F10, mbw, mbh, base.max_intra_level, lw, 10, 1, passes, None) == 0
            wide(7)
            ev = [lib.mi355_event_create() for _ in range(4)]
            lib.mi355_event_record(ev[0], None)
            for i, p in enumerate((1, 2, 4)):
                wide(p)
                lib.mi355_event_record(ev[i + 1], None)
            lib.mi355_sync(None)
            passes = {k: lib.mi355_event_elapsed_ms(ev[i], ev[i + 1]) for i, k in enumerate(("recon_inter", "recon_intra", "deblock"))}
            ms = sum(passes.values())
        finally:
            dev.free()
        v = F10 * mbw * mbh / (ms * 1e-3)
        pts.append({"name": "config2_high10_f%d" % F10, "macroblocks_per_s": v, "frames_per_s": v / (mbw * mbh), "ms_per_step": ms, "frames_per_step": F10,
                    "bytes_per_mb_fused": B_FUSED_HIGH10, "fused_fraction_of_hbm_roofline": v * B_FUSED_HIGH10 / HBM_PEAK, "pass_ms": passes,
                    "verified_by": "tests/test_frame_gpu.py::test_config2_high10_full_size_matches_the_frame_checker (same generator, every sample, against the reference's own 10-bit tables)",
                    "note": "config 2 as a High 10 batch: 16-bit samples, 32-bit coefficients (DeviceFrames(bit_depth=10)), planes with line strides, through "
                            "mi355_h264_decode_frames_wide_dev — the second kernel set in its first, plain form (one wave per macroblock, per-sample arithmetic, "
                            "one loop-filter launch per anti-diagonal); parity: the generated High 10 / High 4:2:2 streams against the reference decoder"})
    except Exception as e:
        pts.append({"name": "config2_high10", "error": repr(e)})
    return pts


def detile_jobs(lib, dev, fs, F):
    """device array of F mi355_surface_job: picture f's tiled dst -> its own linear planes"""
    class Job(C.Structure):
        _fields_ = [("lin", C.c_void_p * 3), ("tiled", C.c_void_p * 2), ("lin_stride", C.c_int32 * 2), ("tiled_stride", C.c_int32 * 2),
                    ("mb_width", C.c_int32), ("mb_height", C.c_int32), ("to_tiled", C.c_int32), ("reserved0", C.c_int32)]
    lib.mi355_h264_surface_convert_dev.restype = C.c_int
    lib.mi355_h264_surface_convert_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    ysz, csz = fs.H * fs.W, fs.H * fs.W // 4
    lin = dev.alloc(F * (ysz + 2 * csz))
    arr = (Job * F)()
    for f in range(F):
        j, d = arr[f], dev.host_desc[f]
        base = lin + f * (ysz + 2 * csz)
        j.lin[0], j.lin[1], j.lin[2] = base, base + ysz, base + ysz + csz
        j.tiled[0], j.tiled[1] = d.dst[0], d.dst[1]
        j.lin_stride[0], j.lin_stride[1] = fs.W, fs.W // 2
        j.tiled_stride[0], j.tiled_stride[1] = d.dst_stride[0], d.dst_stride[1]
        j.mb_width, j.mb_height, j.to_tiled = fs.mb_w, fs.mb_h, 0
    p = dev.alloc(C.sizeof(arr))
    lib.mi355_memcpy_h2d(p, C.addressof(arr), C.sizeof(arr))
    return p


def session_points(lib):
    """Real streams end to end on the device side: the records the REFERENCE decoder's own run
    exported for realshort.mp4 (36 pictures, 320x240) and for a generated QCIF stream (10 pictures I / P, four slices, four
    references, I_PCM; tests/golden) through whole-frame sessions in ONE group — picture i of every session in one launch set,
    every picture predicted from the surfaces the session decoded before, host-to-device copy of each picture's records
    included; fed by this one Python thread."""
    import time
    import session_cases as SC
    import stream_fixture as SF
    out = []

    class P:
        pass
    prov = P()
    prov.lib = lib
    for name, npz, S in (("sessions_group_realshort_x64", SC.SF_NPZ, 64),
                         ("sessions_group_synth_qcif_x64", os.path.join(ROOT, "tests", "golden", "h264_stream_synth_420_8_qcif.npz"), 64)):
        pics = SF.load_npz(npz)
        nsurf = 8
        grp = SC.Group(lib)
        sess = [SC.Session(lib, pics[0]["mb_w"], pics[0]["mb_h"], nsurf, group=grp) for _ in range(S)]
        try:
            def one_pass():
                for i, pc in enumerate(pics):
                    refs = [s_ % nsurf for s_ in pc["slots"]]
                    mv0, mv1 = pc["mv0"].reshape(-1, 32), pc["mv1"].reshape(-1, 32) if pc["use_l1"] else None
                    for ss in sess:
                        assert ss.start(i % nsurf, refs, pc["use_l1"]) == 0
                        SC.send_picture(ss, pc["mb"], mv0, mv1, pc["coef"], pc["slices"], "runs")
                        assert ss.end() == 0
                    assert grp.flush() == 0
                for ss in sess:
                    ss.get((len(pics) - 1) % nsurf)
            one_pass()
            t0 = time.perf_counter()
            one_pass()
            dt = time.perf_counter() - t0
            n = S * len(pics)
            out.append({"name": name, "sessions": S, "pictures": n, "pictures_per_s": n / dt,
                        "macroblocks_per_s": n * pics[0]["mb_w"] * pics[0]["mb_h"] / dt,
                        "note": "records exported from the reference decoder's run, decoded in sequence on session surfaces, one launch set per "
                                "picture index for all sessions (mi355_h264_group_flush); bit-exactness of this path: tests/test_session_gpu.py"})
        finally:
            for ss in sess:
                ss.close()
            grp.destroy()
    return out


def h264_real_stream_points(lib):
    """What a REAL bitstream sees: the reference's own H.264 decoder, entropy decoding and all, end to end.
    (a) Tier 2 through the bridge (contrib/libav/mi355_h264_bridge.c: reconstruction and loop filter of every picture on the device, the
    decoded-picture buffer in HBM): 64 decoder threads on a generated 1080p I / P / B stream (tests/golden/h264_synth_1080p.samples, 109 KB per
    picture), the SAME binary with everything left to the reference's C functions beside it, on as many threads — the host's entropy decoding is the
    wall on both sides, so this aggregate is the honest figure, not the kernels' synthetic rate.
    (b) Tier 1, the literal per-call boundary `north_star` names (ff_*_init hooks: one synchronous stage / launch / copy-back per DSP call): the hooked
    decoder on realshort.mp4 beside the plain one.  A FUNCTIONAL pin (every shim called by its real caller), not an accelerator — the figure is here so
    that nobody takes it for one."""
    import subprocess
    import struct
    import tempfile
    out = []
    exe = os.path.join(ROOT, "oracle", "_ref", "h264_bridge_gpu")
    for name, fn, what in (("h264_bridge_1080p_x64", "h264_synth_1080p.samples", "generated 1080p stream"),
                           ("h264_bridge_high10_1080p_x64", "h264_synth_1080p_high10.samples", "the same stream as High 10 (10-bit samples: the second kernel set)")):
        src = os.path.join(ROOT, "tests", "golden", fn)
        if not (os.path.exists(exe) and os.path.exists(src)):
            continue
        pt = {"name": name, "what": "reference H.264 decoder + Tier-2 bridge, 64 decoder threads, %s, 3 passes each" % what}
        for key, env in (("bridge", {}), ("reference_c_decoder", {"MI355_BRIDGE_PLAIN": "1"})):
            e = dict(os.environ)
            e.pop("MI355_BRIDGE_PLAIN", None)
            e.update(env)
            r = subprocess.run([exe, src, "-", "64", "3"], capture_output=True, text=True, env=e, timeout=600)
            st = json.loads(r.stdout.strip().splitlines()[-1])
            pt[key] = {k: st[k] for k in ("threads", "pictures_output", "pictures_on_device", "pictures_per_launch_set", "pictures_per_s")}
            pt[key]["macroblocks_per_s"] = st["pictures_per_s"] * 8160
        out.append(pt)
    exe1 = os.path.join(ROOT, "oracle", "_ref", "h264_tier1_gpu")
    clip = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"
    if os.path.exists(exe1) and os.path.exists(clip):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import mp4_samples
        avcc, samples = mp4_samples.extract(clip)
        with tempfile.TemporaryDirectory() as tmp:
            s_path = os.path.join(tmp, "s")
            with open(s_path, "wb") as f:
                f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
                for smp in samples:
                    f.write(struct.pack("<I", len(smp)) + smp)
            pt = {"name": "tier1_hooked_decoder_realshort", "what": "FUNCTIONAL BOUNDARY, not an accelerator: the reference decoder with its DSP tables hooked "
                  "(one synchronous launch per DSP call), 36 pictures of 320x240, process start and device initialisation included"}
            for key, env in (("hooked", {}), ("reference_c_decoder", {"MI355_TIER1_PLAIN": "1"})):
                e = dict(os.environ)
                e.pop("MI355_TIER1_PLAIN", None)
                e.update(env)
                t0 = time.perf_counter()
                r = subprocess.run([exe1, s_path, os.path.join(tmp, "o.yuv")], capture_output=True, text=True, env=e, timeout=900)
                dt = time.perf_counter() - t0
                pt[key] = {"pictures": len(samples), "seconds": dt, "pictures_per_s": len(samples) / dt, "rc": r.returncode}
            out.append(pt)
    return out


def hevc_bridge_points(lib):
    """The reference's own HEVC decoder with the Tier-2 bridge (contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c: every
    prediction block, transform unit and intra block of a picture recorded and run on the device level by level, in-loop filters on the
    same device picture, references in HBM) on two generated streams, and the SAME binary with everything forwarded to the reference's
    C functions beside it (oracle/_ref/hevc_bridge_gpu, built where /root/reference exists).  One decoder, one picture per launch set (the first 1080p stream also with 4 and 16 decoders in the process):
    generated 1920x1080 P / B streams (CTB 64) with 2 % and with 30 % intra coding units outside the first picture (a picture's launches
    follow its dependency levels, and those follow its intra blocks: 4x4 intra blocks everywhere are a wavefront of ~850 levels at 1080p),
    832x480, and two of the small ones (136x72: the launch-bound end of the path)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "hevc_bridge_gpu")
    out = []
    if not os.path.exists(exe):
        return out
    for name in ("pb_1080p_few_intra", "pb_1080p_ctb64", "pb_480p_ctb64", "pb_ctb64_depth0", "i_ctb64"):
        src = os.path.join(ROOT, "tests", "golden", "hevc_synth_%s.samples" % name)
        pt = {"name": "hevc_bridge_" + name}
        # "bridge": the product's default policy (pictures below 1.5 M luma samples stay with the C path: MI355_HEVC_BRIDGE_MIN_PIXELS);
        # "bridge_forced": every picture on the device whatever its size (what round 3 reported as "bridge")
        for key, env in (("bridge", {}), ("bridge_forced", {"MI355_HEVC_BRIDGE_MIN_PIXELS": "0"}),
                         ("bridge_random_access_pictures_on_host", {"MI355_HEVC_BRIDGE_IRAP_ON_HOST": "1"}),
                         ("reference_c_decoder", {"MI355_HEVC_RECON_PLAIN": "1", "MI355_HEVC_LF_PLAIN": "1"})):
            if key == "bridge_random_access_pictures_on_host" and not name.startswith("pb_1080p"):
                continue
            if key == "bridge_forced" and name.startswith("pb_1080p"):
                continue                                        # the policy takes 1080p anyway
            e = dict(os.environ)
            for k in ("MI355_HEVC_RECON_PLAIN", "MI355_HEVC_LF_PLAIN", "MI355_HEVC_BRIDGE_IRAP_ON_HOST", "MI355_HEVC_BRIDGE_MIN_PIXELS"):
                e.pop(k, None)
            e.update(env)
            r = subprocess.run([exe, src, "-", "20"], capture_output=True, text=True, env=e, timeout=600)
            st = json.loads(r.stdout.strip().splitlines()[-1])
            pt[key] = {k: st[k] for k in ("pictures_output", "pictures_reconstructed_on_device", "reconstruction_launches", "dependency_levels", "pictures_per_s")}
        if name == "pb_1080p_few_intra":
            # many decoders in ONE process (a thread each, 4 passes each): their waiting pictures share launch sets, up to four sets side by side on their own
            # streams (commit_launches); "_own_launches": every picture its own launches on the default stream (the bridge before); the C decoder with as many threads
            for nthr in (4, 16):
                for key, env in (("bridge_x%d" % nthr, {}), ("bridge_x%d_own_launches" % nthr, {"MI355_HEVC_BRIDGE_SOLO": "1", "MI355_HEVC_BRIDGE_DEFAULT_STREAM": "1"}),
                                 ("reference_c_decoder_x%d" % nthr, {"MI355_HEVC_RECON_PLAIN": "1", "MI355_HEVC_LF_PLAIN": "1"})):
                    e = dict(os.environ)
                    for k in ("MI355_HEVC_RECON_PLAIN", "MI355_HEVC_LF_PLAIN", "MI355_HEVC_BRIDGE_IRAP_ON_HOST", "MI355_HEVC_BRIDGE_MIN_PIXELS", "MI355_HEVC_BRIDGE_SOLO", "MI355_HEVC_BRIDGE_DEFAULT_STREAM", "MI355_HEVC_BRIDGE_SETS_IN_FLIGHT"):
                        e.pop(k, None)
                    e.update(env)
                    r = subprocess.run([exe, src, "-", "4", str(nthr)], capture_output=True, text=True, env=e, timeout=900)
                    st = json.loads(r.stdout.strip().splitlines()[-1])
                    pt[key] = {k: st[k] for k in ("threads", "outputs_identical", "pictures_output", "pictures_reconstructed_on_device", "reconstruction_launches",
                                                  "pictures_per_launch_set", "pictures_per_s")}
        pt["note"] = ("20 passes over the stream in one process; bit-exactness of this path: tests/test_hevc_bridge_gpu.py (all generated streams); "
                      "bridge_random_access_pictures_on_host: MI355_HEVC_BRIDGE_IRAP_ON_HOST=1, the all-intra first picture of each pass reconstructed by "
                      "the reference's functions on the host (filtered on the device, uploaded once)")
        out.append(pt)
    return out


def hevc_point(lib):
    """BASELINE config 3 (HEVC 10-bit 2160p) as one back-to-back chain on 64 pictures: tools/hevc_chain.py"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hevc_chain
    pt = hevc_chain.measure(lib, pictures=64, steps=3)
    # the same 64 pictures as two chains of 32 on their own streams (one chain's memory-bound stages beside the other's arithmetic): the point's rate; the one-chain form beside it
    two = hevc_chain.measure_pipelines(lib, pictures=64, pipelines=2, steps=6)
    pt["one_pipeline"] = {"ms_per_step": pt["ms_per_step"], "fraction_of_hbm_roofline": pt["fraction_of_hbm_roofline"]}
    if two["ms_per_step"] < pt["ms_per_step"]:
        k = pt["ms_per_step"] / two["ms_per_step"]
        pt["pipelines"] = 2
        pt["ms_per_step"] = two["ms_per_step"]
        for key in ("pictures_per_s", "ctb_per_s", "macroblock_equivalents_per_s"):
            pt[key] *= k
        pt["fraction_of_hbm_roofline"] = two["fraction_of_hbm_roofline"]
    return pt


def sws_points(lib):
    """BASELINE config 5 (libswscale yuv420p -> rgb24): 32 device-resident pictures per launch, the reference's own libswscale
    (oracle/_ref/libswsref.so, C paths) on every logical CPU beside it."""
    import numpy as np
    import sws_support as S
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    out = []
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libswsref.so")
    ref = C.CDLL(ref_path) if os.path.exists(ref_path) else None
    if ref is not None:
        ref.sws_getContext.restype = C.c_void_p
        ref.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
        ref.sws_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    for name, what in (("hd_special", "1920x1080 unscaled, special converter"), ("hd_generic", "1920x1080 unscaled, accurate_rnd generic path"),
                       ("uhd_to_hd", "3840x2160 -> 1920x1080 bicubic")):
        ctx = S.load_context(name)
        d = ctx.desc
        pics = [S.picture(name, seed=s) for s in (1, 2)]
        batch = S.DeviceBatch(lib, ctx, pics, 32)
        try:
            for _ in range(2):
                batch.run()
            lib.mi355_sync(None)
            e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
            lib.mi355_event_record(C.c_void_p(e0), None)
            for _ in range(10):
                batch.run()
            lib.mi355_event_record(C.c_void_p(e1), None)
            lib.mi355_sync(None)
            ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)) / 10
        finally:
            batch.close()
        bytes_frame = d.srcW * d.srcH + 2 * d.chrSrcW * d.chrSrcH + d.dstW * d.dstH * 3
        fps = 32 / (ms * 1e-3)
        pt = {"name": "config5_sws_" + name, "what": what, "frames_per_launch": 32, "ms_per_launch": ms, "frames_per_s": fps,
              "algorithmic_bytes_per_frame": bytes_frame, "fraction_of_hbm_roofline": fps * bytes_frame / HBM_PEAK}
        if ref is not None:
            sw, sh, dw, dh, bic, acc, bitexact = S.CONFIGS[name]
            flags = ref.ref_sws_flags_word(bic, acc, bitexact)
            counts = [0] * len(cpus)
            stop = time.perf_counter() + 2.0

            def work(t):
                try:
                    os.sched_setaffinity(0, {cpus[t]})
                except (AttributeError, OSError):
                    pass
                c = ref.sws_getContext(sw, sh, ref.ref_pix_fmt(0), dw, dh, ref.ref_pix_fmt(1), flags, None, None, None)
                planes = pics[t % 2]
                o = np.zeros((dh, dw * 3), np.uint8)
                src = (C.c_void_p * 4)(*[p.ctypes.data for p in planes], None)
                strides = (C.c_int * 4)(*[p.strides[0] for p in planes], 0)
                dst = (C.c_void_p * 4)(o.ctypes.data, None, None, None)
                dstrides = (C.c_int * 4)(o.strides[0], 0, 0, 0)
                while time.perf_counter() < stop:
                    ref.sws_scale(c, src, strides, 0, sh, dst, dstrides)
                    counts[t] += 1
                ref.sws_freeContext(C.c_void_p(c))
            ths = [threading.Thread(target=work, args=(t,)) for t in range(len(cpus))]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            pt["cpu_baseline"] = {"value": sum(counts) / (time.perf_counter() - t0), "unit": "frames/s", "cores": len(cpus), "kind": "reference",
                                  "sample": "sws_scale of the reference's libswscale (C paths, oracle/_ref/libswsref.so), one context and one picture "
                                            "per pinned thread, ~2 s"}
        out.append(pt)
    return out


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unreadable"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(fs, seconds):
    """The reference's own C functions (oracle/_ref/libref.so, kind "reference"; the scalar oracle, kind "port", where that
    object is missing) on every hardware thread of this box, one pinned thread per logical CPU, each decoding its own
    picture of the same workload repeatedly for ~`seconds`."""
    import providers
    import h264_frames as HF
    orc = providers.oracle()
    lib = orc.lib
    lib.oracle_h264_recon_frame.restype = None
    lib.oracle_h264_deblock_frame.restype = None
    # With oracle/_ref/libref.so present (the reference's own C files compiled where they lie, by oracle/Makefile; the
    # built object travels with the tree), the per-macroblock driver calls the reference's compiled functions.
    kind, what = "port", "the scalar C oracle"
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    if os.path.exists(ref_path):
        try:
            ref = C.CDLL(ref_path)
            fns = [C.cast(getattr(ref, n), C.c_void_p) for n in ("ff_h264dsp_init", "ff_h264qpel_init", "ff_h264chroma_init", "ff_h264_pred_init")]
            lib.oracle_h264frame_bind_tables.restype = None
            lib.oracle_h264frame_bind_tables(*fns)
            kind, what = "reference", "the reference's own C functions (oracle/_ref/libref.so: ff_h264dsp/qpel/chroma/pred_init tables, " \
                                      "--disable-asm equivalent) called by the restated per-macroblock driver"
        except (OSError, AttributeError):
            pass
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    nthreads = max(1, len(cpus))
    nmb = fs.mb_w * fs.mb_h
    # one core first (also warms the tables)
    recon, dst = fs.planes(), fs.planes()
    arr, _ = HF.host_frames(fs, recon, dst)
    t0 = time.perf_counter()
    lib.oracle_h264_recon_frame(C.byref(arr[0]))
    lib.oracle_h264_deblock_frame(C.byref(arr[0]))
    one = time.perf_counter() - t0
    # per-thread private output surfaces; inputs are shared read-only.  The threads are pthreads inside the oracle
    # library (oracle_h264_bench_threads), pinned one per logical CPU: no interpreter in the timed loop.
    import numpy as np
    frames = (HF.Frame * nthreads)()
    keep = []
    for t in range(nthreads):
        C.memmove(C.byref(frames, t * C.sizeof(HF.Frame)), C.byref(arr[t % fs.F]), C.sizeof(HF.Frame))
        bufs = [np.zeros((2, fs.H >> (p > 0), fs.W >> (p > 0)), np.uint8) for p in range(3)]      # [recon, dst] per plane
        keep.append(bufs)
        for p in range(3):
            frames[t].recon[p] = bufs[p][0].ctypes.data
            frames[t].dst[p] = bufs[p][1].ctypes.data
    cpu_arr = (C.c_int * nthreads)(*cpus)
    wall = C.c_double(0.0)
    lib.oracle_h264_bench_threads.restype = C.c_long
    lib.oracle_h264_bench_threads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
    n = lib.oracle_h264_bench_threads(C.cast(frames, C.c_void_p), nthreads, C.cast(cpu_arr, C.c_void_p), float(seconds), C.byref(wall))
    value = n * nmb / wall.value
    return {"value": value, "unit": "macroblocks/s", "cores": nthreads, "kind": kind,
            "value_1core": nmb / one,
            # what the box actually delivered: the threads' aggregate over ONE thread running alone (a leased box is usually
            # CPU-quota'd or shared: 256 pinned threads have measured 9x one core), and the cgroup's own quota where it states one
            "effective_cores": round(value / (nmb / one), 1), "cgroup_cpu_quota_cores": cgroup_cpu_quota(),
            "sample": "%d pictures of the same workload decoded repeatedly for %.0f s on %d pinned pthreads (every logical CPU "
                      "this process may run on); %s (reference x86 SIMD not built: no nasm in the image)"
                      % (min(nthreads, fs.F), seconds, nthreads, what)}


if __name__ == "__main__":
    main()
