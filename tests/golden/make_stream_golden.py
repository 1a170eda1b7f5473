#!/usr/bin/env python3
"""Regenerate tests/golden/h264_stream_realshort.npz (needs /root/reference and the sample clip).

Pipeline: the H.264 track of an .mp4 -> samples (mp4_samples.py) -> the REFERENCE's own decoder,
built in place by oracle/Makefile and instrumented with two linker wraps
(oracle/ref_h264_export.c) -> per-MB Tier-2 records + the reference's decoded pictures.
The clip ships with the imageio Python package of this image (no network needed):
realshort.mp4, H.264 High, 320x240 4:2:0, CABAC, 36 pictures (I/P, 8x8 transform, skip, intra-in-P).
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
CLIP = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"


def main():
    import mp4_samples
    import stream_fixture as SF
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_export"], check=True)
    with tempfile.TemporaryDirectory() as d:
        avcc, samples = mp4_samples.extract(CLIP)
        import struct
        with open(os.path.join(d, "s"), "wb") as f:
            f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
            for s in samples:
                f.write(struct.pack("<I", len(s)) + s)
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "h264_export"), os.path.join(d, "s"), os.path.join(d, "o")], check=True)
        pics = SF.parse_export(os.path.join(d, "o"))
    out = os.path.join(HERE, "h264_stream_realshort.npz")
    SF.save_npz(out, pics)
    print("wrote", out, os.path.getsize(out), "bytes,", len(pics), "pictures")


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
