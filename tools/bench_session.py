#!/usr/bin/env python3
"""Developer measurement: pictures/s through whole-frame sessions (include/mi355_h264_session.h) — one picture per launch
set, host -> device copy of the picture's records included.  S sessions are fed round-robin from one thread (their HIP
streams overlap on the device).  GPU box: python tools/bench_session.py [--size 1080p|realshort] [--sessions 1,4,16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np

import h264_frames as HF
import providers
import session_cases as SC
import stream_fixture as SF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080p")
    ap.add_argument("--sessions", default="1,4,16")
    ap.add_argument("--pictures", type=int, default=64)
    ap.add_argument("--group", action="store_true", help="all sessions in one group: one launch set per step for all of them")
    a = ap.parse_args()
    prov = providers.mi355()
    if a.size == "realshort":
        pics = SF.load_npz(SC.SF_NPZ)
        mb_w, mb_h = pics[0]["mb_w"], pics[0]["mb_h"]
        seq = [(pc["mb"], pc["mv0"].reshape(-1, 32), pc["coef"], pc["slices"], [s_ for s_ in pc["slots"]]) for pc in pics]
    else:
        fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=prov.lib, nrefs=1)
        mb_w, mb_h = 120, 68
        seq = [(fs.mb[f], fs.mv[0, f].reshape(-1, 32), fs.coef[f], fs.slices[f], [f - 1] if f else []) for f in range(4)]
    for S in [int(x) for x in a.sessions.split(",")]:
        grp = SC.Group(prov.lib) if a.group else None
        sess = [SC.Session(prov.lib, mb_w, mb_h, 3, group=grp) for _ in range(S)]
        try:
            if a.size != "realshort":
                gray = [np.full((16 * mb_h, 16 * mb_w), 128, np.uint8), np.full((8 * mb_h, 8 * mb_w), 128, np.uint8), np.full((8 * mb_h, 8 * mb_w), 128, np.uint8)]
                for ss in sess:
                    ss.put(2, gray)

            def one_pass(n):
                # picture i goes to surface i % 3 and is predicted from picture i - 1 (realshort: its I picture has no reference)
                for i in range(n):
                    mb, mv0, coef, sl, refs = seq[i % len(seq)]
                    r = [(i - 1) % 3] if (len(refs) or a.size != "realshort") else []
                    for ss in sess:
                        assert ss.start(i % 3, r, False) == 0
                        assert ss.slice(sl[:1], 0, mb, mv0, None, coef) == 0
                        assert ss.end() == 0
                    if grp:
                        assert grp.flush() == 0
                for ss in sess:
                    ss.get((n - 1) % 3)
            one_pass(4)
            t0 = time.perf_counter()
            one_pass(a.pictures)
            dt = time.perf_counter() - t0
            print(json.dumps({"size": a.size, "grouped": bool(a.group), "sessions": S, "pictures": a.pictures * S, "pictures_per_s": round(a.pictures * S / dt, 1),
                              "mb_per_s": round(a.pictures * S * mb_w * mb_h / dt)}), flush=True)
        finally:
            for ss in sess:
                ss.close()
            if grp:
                grp.destroy()


if __name__ == "__main__":
    main()
