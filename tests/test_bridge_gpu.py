"""GPU: the reference's H.264 decoder with the Tier-2 bridge (contrib/libav/mi355_h264_bridge.c) bound to the real HIP
library (oracle/_ref/h264_bridge_gpu, built where /root/reference exists and shipped with the tree):
 * realshort.mp4: every picture equals the unmodified reference decoder's (synchronous and lazy completion, 1 and 4
   decoder threads = streams, batched through the dispatcher and direct);
 * cockatoo.mp4 (High 4:4:4 Predictive 720p, P + B pictures, explicit and implicit weighted prediction): all 280 pictures
   through the device (three plane passes per picture) equal the plain run of the same binary."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_bridge_emu import CLIP, ROOT, check_against_golden, samples_file

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "oracle", "_ref", "h264_bridge_gpu")
CLIP444 = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/cockatoo.mp4"


def _run(args, env_extra=None, timeout=600):
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/h264_bridge_gpu missing: run __graft_entry__.build() where /root/reference exists")
    env = dict(os.environ)
    for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_PLAIN", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_SESSION", "MI355_BRIDGE_LINEAR"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


@pytest.mark.parametrize("lazy,direct,threads", ((False, False, 1), (True, False, 1), (False, False, 6), (True, False, 6), (False, True, 1), (True, True, 4),
                                                 (False, "session", 1), (False, "session", 4), (False, "linear", 3)))
def test_bridge_decodes_realshort_on_gpu(tmp_path, mi355, lazy, direct, threads):
    """batched / lazy / direct submission; "session": every picture through the whole-frame session façade
    (MI355_BRIDGE_SESSION: mi355_h264_start_frame / decode_slice / end_frame called by the reference decoder's bridge);
    "linear": device pictures as planes with line strides instead of macroblock tiles"""
    if not os.path.exists(CLIP):
        pytest.skip("sample clip not in this image")
    src, n = samples_file(tmp_path, CLIP)
    out = tmp_path / "o.yuv"
    env = {}
    if lazy:
        env["MI355_BRIDGE_LAZY"] = "1"
    if direct == "session":
        env["MI355_BRIDGE_SESSION"] = "1"
    elif direct == "linear":
        env["MI355_BRIDGE_LINEAR"] = "1"
        direct = False
    elif direct:
        env["MI355_BRIDGE_DIRECT"] = "1"
    direct = bool(direct)
    stats, err = _run([src, out, threads, 2], env)
    assert stats["pictures_output"] == 2 * n * threads and stats["pictures_on_device"] == 2 * n * threads and stats["bridges_active"] == threads, (stats, err[-500:])
    assert (stats["launch_sets"] == 0) == direct
    check_against_golden(np.fromfile(out, np.uint8), n)


@pytest.mark.parametrize("mode", ("batched", "lazy", "direct", "threads3"))
def test_bridge_decodes_444_clip_on_gpu(tmp_path, mi355, mode):
    """cockatoo.mp4: High 4:4:4 Predictive 1280x720, 280 pictures, CABAC, P pictures with explicit weights and B pictures
    with implicit weights (weighted_bipred_idc 2), 4 reference frames: every plane goes through the device as its own pass
    (the plane in the luma role).  Every output picture must equal the plain run of the same binary (MI355_BRIDGE_PLAIN:
    the reference's own C path)."""
    if not os.path.exists(CLIP444):
        pytest.skip("sample clip not in this image")
    n = 280 if mode == "batched" else 60
    src, n = samples_file(tmp_path, CLIP444, n=n)
    a, b = tmp_path / "a.yuv", tmp_path / "b.yuv"
    env = {"lazy": {"MI355_BRIDGE_LAZY": "1"}, "direct": {"MI355_BRIDGE_DIRECT": "1"}}.get(mode, {})
    threads = 3 if mode == "threads3" else 1
    stats, err = _run([src, a, threads, 1], env)
    assert stats["bridges_active"] == threads and stats["pictures_on_device"] == n * threads and stats["pictures_output"] == n * threads, (stats, err[-500:])
    plain, _ = _run([src, b, 1, 1], {"MI355_BRIDGE_PLAIN": "1"})
    assert plain["pictures_on_device"] == 0 and plain["pictures_output"] == n
    got, want = np.fromfile(a, np.uint8), np.fromfile(b, np.uint8)
    assert got.size == want.size == n * 1280 * 720 * 3
    if not np.array_equal(got, want):
        fs = 1280 * 720 * 3
        bad = [i for i in range(n) if not np.array_equal(got[i * fs:(i + 1) * fs], want[i * fs:(i + 1) * fs])]
        raise AssertionError("pictures differ from the reference decoder's: %s" % bad[:10])


def test_bridge_mixed_streams_on_gpu(tmp_path, mi355):
    """two streams of different size and chroma format (320x240 4:2:0 and 1280x720 4:4:4) decoded at the same time by four
    threads: their pictures meet in the dispatcher's launch sets (mixed geometry in one batch); both outputs must equal
    the plain runs"""
    if not (os.path.exists(CLIP) and os.path.exists(CLIP444)):
        pytest.skip("sample clips not in this image")
    d1, d2 = tmp_path / "x", tmp_path / "y"
    d1.mkdir(); d2.mkdir()
    s1, n1 = samples_file(d1, CLIP)
    s2, n2 = samples_file(d2, CLIP444, n=36)
    out = tmp_path / "o.yuv"
    stats, err = _run(["%s,%s" % (s1, s2), out, 4, 2])
    assert stats["bridges_active"] == 4 and stats["pictures_on_device"] == 2 * 2 * (n1 + n2), (stats, err[-500:])
    check_against_golden(np.fromfile(out, np.uint8), n1)
    ref = tmp_path / "r.yuv"
    _run([s2, ref, 1, 1], {"MI355_BRIDGE_PLAIN": "1"})
    assert np.array_equal(np.fromfile(str(out) + ".1", np.uint8), np.fromfile(ref, np.uint8))
