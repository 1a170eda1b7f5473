"""GPU: the REFERENCE's own H.264 decoder, its five DSP tables overridden through the linker by
this project's ff_*_init_mi355x hooks and bound to the real HIP library, decodes real clips on the MI355X.
oracle/_ref/h264_tier1_gpu is built where /root/reference exists (oracle/Makefile, __graft_entry__.build()) and travels
to the GPU box with the tree; the clips come with the image's imageio package.
 * realshort.mp4 (High 4:2:0, 36 pictures): every picture equals what the unmodified reference decoder produced
   (tests/golden/h264_stream_realshort.npz).
 * cockatoo.mp4 (High 4:4:4 Predictive 1280x720, 280 pictures): hooked vs plain run of the same binary.  The first 24
   pictures by default (Tier 1 is one synchronous launch per DSP call); MI355_444_PICTURES=280 runs the whole clip
   (198 s on the GPU box, identical: profiles/r02_tier1_decoder_gpu.txt)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "h264_tier1_gpu")
CLIP = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"
CLIP444 = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/cockatoo.mp4"


def _samples_file(tmp_path, clip, n=None):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mp4_samples
    avcc, samples = mp4_samples.extract(clip)
    n = len(samples) if n is None else n
    src = tmp_path / "s"
    with open(src, "wb") as f:
        f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", n))
        for s in samples[:n]:
            f.write(struct.pack("<I", len(s)) + s)
    return src, n


def _need():
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/h264_tier1_gpu missing: run __graft_entry__.build() where /root/reference exists")


def test_reference_decoder_on_gpu_realshort(tmp_path, mi355):
    _need()
    if not os.path.exists(CLIP):
        pytest.skip("sample clip not in this image")
    import stream_fixture as SF
    src, n = _samples_file(tmp_path, CLIP)
    out = tmp_path / "o.yuv"
    r = subprocess.run([EXE, str(src), str(out)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    pics = SF.load_npz(os.path.join(ROOT, "tests", "golden", "h264_stream_realshort.npz"))
    raw = np.fromfile(out, np.uint8)
    w, h = 16 * pics[0]["mb_w"], 16 * pics[0]["mb_h"]
    fsz = w * h * 3 // 2
    assert raw.size == n * fsz, (raw.size, n, fsz, r.stderr[-300:])
    for i in range(n):
        want = np.concatenate([pics[i][k].reshape(-1) for k in ("y", "cb", "cr")])
        assert np.array_equal(raw[i * fsz:(i + 1) * fsz], want), "picture %d differs from the reference decoder's" % i


def test_reference_decoder_on_gpu_444_clip(tmp_path, mi355):
    _need()
    if not os.path.exists(CLIP444):
        pytest.skip("sample clip not in this image")
    src, n = _samples_file(tmp_path, CLIP444, n=int(os.environ.get("MI355_444_PICTURES", "24")))
    outs = []
    for plain in (True, False):
        out = tmp_path / ("plain.yuv" if plain else "hooked.yuv")
        env = dict(os.environ)
        env.pop("MI355_TIER1_PLAIN", None)
        if plain:
            env["MI355_TIER1_PLAIN"] = "1"
        r = subprocess.run([EXE, str(src), str(out)], capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.fromfile(out, np.uint8))
    assert outs[0].size == outs[1].size and outs[0].size >= 1280 * 720 * 3, (outs[0].size, outs[1].size)
    assert np.array_equal(outs[0], outs[1]), "4:4:4 pictures differ: %d samples" % int((outs[0] != outs[1]).sum())
