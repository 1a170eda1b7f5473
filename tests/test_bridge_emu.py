"""CPU, only where /root/reference exists: the reference's H.264 decoder with the Tier-2 BRIDGE
(contrib/libav/mi355_h264_bridge.c — product code: the host only parses; reconstruction and loop filter are the
batched kernels) bound to the SIMT-emulated build of the product sources, decodes a real clip: every output picture
must equal what the unmodified reference decoder produced."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"


def samples_file(tmp_path, clip, n=None):
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


def check_against_golden(raw, n):
    import stream_fixture as SF
    pics = SF.load_npz(os.path.join(ROOT, "tests", "golden", "h264_stream_realshort.npz"))
    # the harness writes the CROPPED pictures the decoder outputs; the fixture holds the coded size (here the same)
    w, h = 16 * pics[0]["mb_w"], 16 * pics[0]["mb_h"]
    fsz = w * h * 3 // 2
    assert raw.size == n * fsz, (raw.size, n, fsz)
    for i in range(n):
        want = np.concatenate([pics[i][k].reshape(-1) for k in ("y", "cb", "cr")])
        assert np.array_equal(raw[i * fsz:(i + 1) * fsz], want), "picture %d differs from the reference decoder's" % i


@pytest.mark.skipif(not (os.path.isdir("/root/reference/libavcodec") and os.path.exists(CLIP)), reason="needs /root/reference and the sample clip")
@pytest.mark.parametrize("lazy", (False, True))
def test_bridge_decodes_realshort_on_the_emulator(tmp_path, emu, lazy):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    src, n = samples_file(tmp_path, CLIP)
    out = tmp_path / "o.yuv"
    env = dict(os.environ)
    env.pop("MI355_BRIDGE_LAZY", None)
    if lazy:
        env["MI355_BRIDGE_LAZY"] = "1"
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "h264_bridge_emu"), str(src), str(out)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["pictures_output"] == n and stats["pictures_on_device"] == n and stats["bridges_active"] == 1, (stats, r.stderr[-500:])
    check_against_golden(np.fromfile(out, np.uint8), n)
