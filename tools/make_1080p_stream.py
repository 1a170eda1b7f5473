#!/usr/bin/env python3
"""Developer tool (needs /root/reference: the CAVLC code tables are read from its source): the generated 1080p 4:2:0 stream
tools/bridge_1080p.sh decodes — 120 x 68 macroblocks, 10 pictures I / P / B (implicit weights), four slices, 8x8 transform,
three references, sparse residuals and 45 % skipped macroblocks: 109 KB per picture.  -> tests/golden/h264_synth_1080p.samples, and the same
parameters at 10 bits -> h264_synth_1080p_high10.samples (both committed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_h264_streams as M

T = M.load_tables()
kw = dict(mb_w=120, mb_h=68, chroma_idc=1, depth=8, seed=2024, nslices=4, deblock_idc=0, nrefs=3, npics=10, bmode=1, t8x8=True, far=24, sparse=0.35, skip=0.45)
out = os.path.join(ROOT, "tests", "golden")        # committed (1.1 MB each): bench.py's h264_bridge_1080p points and tools/bridge_1080p.sh decode them
for name, extra in (("h264_synth_1080p.samples", {}), ("h264_synth_1080p_high10.samples", dict(depth=10))):      # the same stream as High 10
    units = M.Stream(T, "hd", **dict(kw, **extra)).build()
    M.write_samples(os.path.join(out, name), units)
    print(name, len(units), "pictures", sum(map(len, units)), "bytes")
