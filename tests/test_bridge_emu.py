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
@pytest.mark.parametrize("lazy,direct,threads,linear", ((False, False, 1, False), (True, False, 1, False), (False, True, 1, False), (True, True, 1, False), (False, False, 3, False),
                                                        (False, False, 1, True), (True, True, 1, True), (False, False, 1, "session"), (False, False, 1, "device1")))
def test_bridge_decodes_realshort_on_the_emulator(tmp_path, emu, lazy, direct, threads, linear):
    """batched submission through the dispatcher thread (default) and direct submission (MI355_BRIDGE_DIRECT), complete
    at once or lazily; with 3 decoder threads the dispatcher's launch sets hold pictures of several streams.  The default
    form decodes the whole clip, the others its first 12 pictures (I, P and B pictures; the emulator is slow)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    src, n = samples_file(tmp_path, CLIP, None if (lazy, direct, threads, linear) == (False, False, 1, False) else 12)
    out = tmp_path / "o.yuv"
    env = dict(os.environ)
    for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_PLAIN", "MI355_BRIDGE_LINEAR", "MI355_BRIDGE_SESSION"):
        env.pop(k, None)
    if linear == "device1":                       # the second of two (emulated) GPUs: the decoder thread, its dispatcher and its pictures live there
        env["MI355_EMU_DEVICES"], env["MI355_DEVICE"] = "2", "1"
        linear = False
    if linear == "session":                       # the whole-frame session façade as the submission path
        env["MI355_BRIDGE_SESSION"] = "1"
        linear, direct = False, True
    if lazy:
        env["MI355_BRIDGE_LAZY"] = "1"
    if direct:
        env["MI355_BRIDGE_DIRECT"] = "1"
    if linear:
        env["MI355_BRIDGE_LINEAR"] = "1"          # device pictures as planes with line strides (the default for this clip: macroblock tiles)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "h264_bridge_emu"), str(src), str(out), str(threads), "1"], capture_output=True, text=True,
                       timeout=1800, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["pictures_output"] == n * threads and stats["pictures_on_device"] == n * threads and stats["bridges_active"] == threads, (stats, r.stderr[-500:])
    if direct:
        assert stats["launch_sets"] == 0
    else:
        assert stats["launch_sets"] >= n and stats["launch_sets"] * stats["pictures_per_launch_set"] == pytest.approx(n * threads, rel=0.01)   # the harness prints two decimals
        if threads > 1:
            assert stats["launch_sets"] < n * threads                   # some launch sets held more than one stream's picture
    check_against_golden(np.fromfile(out, np.uint8), n)


def test_bridge_plain_run_is_the_reference_path(tmp_path):
    """MI355_BRIDGE_PLAIN=1: the bridge never touches the device and the decoder's own C path produces the golden pictures"""
    if not (os.path.isdir("/root/reference/libavcodec") and os.path.exists(CLIP)):
        pytest.skip("needs /root/reference and the sample clip")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    src, n = samples_file(tmp_path, CLIP)
    out = tmp_path / "o.yuv"
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "h264_bridge_emu"), str(src), str(out), "1", "1"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, MI355_BRIDGE_PLAIN="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["pictures_output"] == n and stats["pictures_on_device"] == 0 and stats["bridges_active"] == 0 and r.stderr.strip() == ""
    check_against_golden(np.fromfile(out, np.uint8), n)


def test_bridge_survives_a_damaged_stream(tmp_path, emu):
    """a P picture's slice cut to a third of its bytes: whatever the reference's entropy decoder makes of it (with this CABAC
    clip it reads the padding as skipped macroblocks and completes the picture; a decoder that gives up instead leaves the
    picture incomplete, which makes the bridge step aside after bringing back what it was given), the bridged run must
    behave like the plain one: same exit code, same number of pictures, identical pictures before the damage."""
    if not (os.path.isdir("/root/reference/libavcodec") and os.path.exists(CLIP)):
        pytest.skip("needs /root/reference and the sample clip")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import mp4_samples
    avcc, samples = mp4_samples.extract(CLIP)
    bad = 9
    # cut the picture's largest NAL unit (its slice) to a third: the entropy decoder runs out of data in the middle of the picture
    s, pos, units = samples[bad], 0, []
    while pos + 4 <= len(s):
        n = int.from_bytes(s[pos:pos + 4], "big")
        units.append(s[pos + 4:pos + 4 + n])
        pos += 4 + n
    big = max(range(len(units)), key=lambda i: len(units[i]))
    units[big] = units[big][:max(8, len(units[big]) // 3)]
    samples[bad] = b"".join(len(u).to_bytes(4, "big") + u for u in units)
    samples = samples[:bad + 6]                    # a few pictures past the damage (the emulator is slow)
    src = tmp_path / "s"
    with open(src, "wb") as f:
        f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
        for x in samples:
            f.write(struct.pack("<I", len(x)) + x)
    out, ref = tmp_path / "o.yuv", tmp_path / "r.yuv"
    exe = os.path.join(ROOT, "oracle", "_ref", "h264_bridge_emu")
    r = subprocess.run([exe, str(src), str(out), "1", "1"], capture_output=True, text=True, timeout=900)
    p = subprocess.run([exe, str(src), str(ref), "1", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, MI355_BRIDGE_PLAIN="1"))
    assert r.returncode == p.returncode
    got, want = np.fromfile(out, np.uint8), np.fromfile(ref, np.uint8)
    assert got.size == want.size and got.size > 0                                   # same number of pictures as the plain run
    fs = 320 * 240 * 3 // 2
    assert np.array_equal(got[:bad * fs], want[:bad * fs])                          # everything before the damage
    if "incomplete picture" in r.stderr:
        stats = json.loads(r.stdout.strip().splitlines()[-1])
        assert stats["bridges_active"] == 0 and stats["pictures_on_device"] >= bad
