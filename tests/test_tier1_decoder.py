"""CPU, only where /root/reference exists: the REFERENCE's own H.264 decoder, with its five DSP tables
overridden through the linker by this project's ff_*_init_mi355x hooks (here bound to
the SIMT-emulated build of the product sources), decodes a real clip and must reproduce, sample for
sample, what the unmodified reference decoder produced (the pictures stored in the stream fixture).
This exercises every Tier-1 shim under its real caller: offsets, strides, edge emulation, call order."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"


@pytest.mark.skipif(not (os.path.isdir("/root/reference/libavcodec") and os.path.exists(CLIP)),
                    reason="needs /root/reference and the sample clip")
def test_reference_decoder_through_tier1_hooks(tmp_path, emu):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mp4_samples
    import stream_fixture as SF
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_tier1_emu"], check=True)
    avcc, samples = mp4_samples.extract(CLIP)
    n = len(samples)                         # 36 I + P pictures with skip, 8x8 transform and intra-in-P MBs
    src = tmp_path / "s"
    with open(src, "wb") as f:
        f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", n))
        for s in samples[:n]:
            f.write(struct.pack("<I", len(s)) + s)
    out = tmp_path / "o.yuv"
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "h264_tier1_emu"), str(src), str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    pics = SF.load_npz(os.path.join(ROOT, "tests", "golden", "h264_stream_realshort.npz"))
    raw = np.fromfile(out, np.uint8)
    w, h = 16 * pics[0]["mb_w"], 16 * pics[0]["mb_h"]
    fsz = w * h * 3 // 2
    assert raw.size == n * fsz, (raw.size, n, fsz, r.stderr[-300:])
    for i in range(n):
        fr = raw[i * fsz:(i + 1) * fsz]
        want = np.concatenate([pics[i][k].reshape(-1) for k in ("y", "cb", "cr")])
        assert np.array_equal(fr, want), "picture %d differs from the reference decoder's" % i


CLIP444 = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/cockatoo.mp4"


@pytest.mark.skipif(not (os.path.isdir("/root/reference/libavcodec") and os.path.exists(CLIP444)),
                    reason="needs /root/reference and the sample clip")
def test_reference_decoder_444_clip_through_tier1_hooks(tmp_path, emu):
    """High 4:4:4 Predictive 1280x720 (SURVEY.md §8c's second offline clip): the 4:4:4 decode path calls the luma
    entries for all three planes.  First pictures only (the emulator is slow); hooked vs plain run of the same binary."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mp4_samples
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_tier1_emu"], check=True)
    avcc, samples = mp4_samples.extract(CLIP444)
    n = 5
    src = tmp_path / "s"
    with open(src, "wb") as f:
        f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", n))
        for s in samples[:n]:
            f.write(struct.pack("<I", len(s)) + s)
    exe = os.path.join(ROOT, "oracle", "_ref", "h264_tier1_emu")
    outs = []
    for plain in (True, False):
        out = tmp_path / ("plain.yuv" if plain else "hooked.yuv")
        env = dict(os.environ)
        if plain:
            env["MI355_TIER1_PLAIN"] = "1"
        else:
            env.pop("MI355_TIER1_PLAIN", None)
        r = subprocess.run([exe, str(src), str(out)], capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.fromfile(out, np.uint8))
    assert outs[0].size == outs[1].size and outs[0].size >= 1280 * 720 * 3, (outs[0].size, outs[1].size)
    assert np.array_equal(outs[0], outs[1]), "4:4:4 pictures differ: %d samples" % int((outs[0] != outs[1]).sum())
