#!/usr/bin/env python3
"""Throughput of the swscale kernels on SURVEY.md §8d config 5 (not the headline metric; numbers go
to DESIGN.md): batch of device-resident pictures, HIP events around K launches, algorithmic bytes =
source planes read once + RGB24 written once.  Also times the CPU oracle on one picture."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import providers  # noqa: E402
import sws_support as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--configs", default="hd_special,hd_generic,uhd_to_hd")
    a = ap.parse_args()
    prov = providers.mi355()
    lib = prov.lib
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    orc = S.oracle_backend(providers.oracle())
    for name in a.configs.split(","):
        ctx = S.load_context(name)
        d = ctx.desc
        pics = [S.picture(name, seed=s) for s in (1, 2)]
        batch = S.DeviceBatch(lib, ctx, pics, a.frames)
        for _ in range(3):
            batch.run()
        lib.mi355_sync(None)
        e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
        lib.mi355_event_record(C.c_void_p(e0), None)
        for _ in range(a.steps):
            batch.run()
        lib.mi355_event_record(C.c_void_p(e1), None)
        lib.mi355_sync(None)
        ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)) / a.steps
        bytes_frame = d.srcW * d.srcH + 2 * d.chrSrcW * d.chrSrcH + d.dstW * d.dstH * 3
        fps = a.frames / (ms * 1e-3)
        t = time.time()
        orc.scale(ctx, pics[0])
        cpu = time.time() - t
        print(json.dumps({"workload": name, "frames_per_launch": a.frames, "ms_per_launch": ms, "frames_per_s": fps,
                          "algorithmic_bytes_per_frame": bytes_frame, "achieved_GBps": fps * bytes_frame / 1e9,
                          "frac_of_8TBps": fps * bytes_frame / 8e12, "cpu_oracle_frames_per_s_1core": 1.0 / cpu}))
        batch.close()


if __name__ == "__main__":
    main()
