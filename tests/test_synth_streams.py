"""CPU: the generated streams (synth_streams.py) — oracle, emulated kernels, sessions, and (where /root/reference's decoder
harnesses are built) the reference decoder with this project's Tier-1 hooks / Tier-2 bridge on the SIMT emulator."""
import json
import os

import numpy as np
import pytest

import h264_frames as HF
import session_cases as SC
import stream_fixture as SF
import synth_streams as SY

needs_harness = pytest.mark.skipif(not os.path.isdir("/root/reference/libavcodec"), reason="needs the reference decoder objects (/root/reference)")


def _check(pics, planes):
    for f in range(planes[0].shape[0]):
        for p, key in enumerate(("y", "cb", "cr")):
            assert np.array_equal(planes[p][f], pics[f][key]), "picture %d plane %s differs from the reference decoder" % (f, key)


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_fixture_covers_what_the_clips_lack(name):
    pics = SF.load_npz(SY.npz(name))
    assert max(len(p["slices"]) for p in pics) >= 2 or name == "420_8_b_explicit"         # several slices per picture
    assert any((p["mb"]["mb_type"] & HF.I_PCM_BIT).any() for p in pics) if hasattr(HF, "I_PCM_BIT") else True
    assert max(len(p["slots"]) for p in pics) >= 2                                        # more than one reference
    if name not in ("420_8_nofilter", "420_8_b_average", "420_8_cip_mixed"):
        assert any((p["slices"]["use_weight"] != 0).any() for p in pics)                  # explicit / implicit weights, 4:2:0
    if "_b_" in name:
        assert any(p["pict_type"] == 3 and p["use_l1"] for p in pics)                     # B pictures
        assert {"420_8_b_implicit": 2, "420_8_b_explicit": 1, "420_8_b_average": 0}[name] == max(int(p["slices"]["use_weight"].max()) for p in pics if p["pict_type"] == 3)


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_oracle_reproduces_reference_decoder_on_generated_streams(oracle, name):
    pics = SF.load_npz(SY.npz(name))
    _, dst = HF.run_oracle(oracle, SF.frameset_all(pics))
    _check(pics, dst)


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_emulated_kernels_reproduce_reference_decoder_on_generated_streams(emu, name):
    pics = SF.load_npz(SY.npz(name))
    d = HF.DeviceFrames(emu, SF.frameset_all(pics))
    try:
        d.decode()
        _check(pics, d.fetch(d.dst))
    finally:
        d.free()


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_session_decodes_generated_streams_in_sequence_emulated(emu, name):
    """every picture on the session's own surfaces, up to four of them as references; slices as runs / address lists / split"""
    SC.run_stream(emu, SY.npz(name), 0, None, nsurf=8, sync_each=False)


@needs_harness
@pytest.mark.parametrize("name", SY.ALL)
def test_reference_decoder_with_tier1_hooks_emulated(tmp_path, emu, name):
    """8-bit 4:2:0 / 4:2:2, 9- and 10-bit: the reference decoder with its five DSP tables overridden by the hooks (emulated
    kernels) outputs what it outputs with its own tables"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_tier1_emu"], check=True)
    out = tmp_path / "plain.yuv"
    SY.run_tier1("h264_tier1_emu", name, out, plain=True)
    SY.check_md5(out, name)                                      # the committed md5 is the reference's
    out = tmp_path / "hooked.yuv"
    line = SY.run_tier1("h264_tier1_emu", name, out)
    assert SY.MD5[name]["summary"] in line
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("lazy", (False, True))
@pytest.mark.parametrize("name", SY.BRIDGE)
def test_bridge_decodes_generated_streams_emulated(tmp_path, emu, name, lazy):
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, lazy=lazy)
    assert st.get("pictures_on_device") == SY.ON_DEVICE.get(name, SY.MD5[name]["pictures"]), st           # nothing fell back to the C path
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name,no_wide", [(n, False) for n in SY.OUTSIDE] + [(n, True) for n in ("422_8_b", "420_10_t8x8", "444_10", "422_10_paff", "420_8_lossless", "444_8_lossless", "422_10_lossless", "420_8_mbaff", "444_8_mbaff", "422_10_mbaff")])
def test_bridge_steps_aside_for_streams_outside_tier2(tmp_path, emu, name, no_wide):
    """High 4:2:2, 9 / 10 bit, transform bypass and MBAFF with the second kernel set switched off (MI355_BRIDGE_NO_WIDE): the bridge says so once and the
    reference's C path decodes the stream — same pictures, nothing on the device"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, no_wide=no_wide)
    assert st.get("pictures_on_device") == 0 and st.get("pictures_output") == SY.MD5[name]["pictures"], st
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name,on_device", (("420_8_2wide_b", 0), ("420_8_paff_idc2_intra", None)))
def test_bridge_leaves_the_reference_s_inconsistent_cases_to_it(tmp_path, emu, name, on_device):
    """pictures two macroblocks wide with two-reference weighted prediction (h264_mb.c:407-409), field pictures with disable_deblocking_filter_idc 2
    (h264_mb.c:525-527): the reference's output there follows from its own buffers, not from the standard — the bridge says so once and hands the decoder back
    (the whole stream / from the first such field picture on); the output is the reference's"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, keep_field_idc2=False)
    assert st.get("pictures_output") == SY.MD5[name]["pictures"], st
    if on_device is None:
        assert 0 < st.get("pictures_on_device") < SY.MD5[name]["pictures"], st        # the frame pictures before the first field picture were the device's
    else:
        assert st.get("pictures_on_device") == on_device, st
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name,on_device,frames", (("mixed_formats", 12, 12), ("paff_and_frames", 23, 17)))
@pytest.mark.parametrize("lazy,direct", ((False, False), (True, False), (False, True)))
def test_bridge_follows_sequence_changes_emulated(tmp_path, emu, lazy, direct, name, on_device, frames):
    """a stream whose sequences differ in chroma format and bit depth (8-bit 4:2:0, 10-bit 4:2:2, 8-bit 4:4:4, 8-bit 4:2:0): the
    bridge gives its buffers back at each change (the decoder calls ff_h264_flush_change) and sets itself up for the next format —
    the 10-bit 4:2:2 sequence goes through the second kernel set; `paff_and_frames`: PAFF 4:2:0, progressive with B pictures at the
    same size (buffers kept), PAFF 4:4:4, PAFF 4:2:2; `420_8_resize` (three picture sizes, all on the device) is among SY.BRIDGE"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, lazy=lazy, direct=direct)
    assert st.get("pictures_on_device") == on_device and st.get("pictures_output") == frames, st
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name,on_device", (("420_8_resize", 11), ("mixed_formats", 12)))
def test_bridge_sequence_changes_with_several_decoders_emulated(tmp_path, emu, name, on_device):
    """four decoder threads, each decoding the stream twice: buffers are given back and set up again while the other decoders'
    pictures are in the dispatcher's launch sets"""
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, threads=4, loops=2, lazy=True)
    assert st.get("pictures_on_device") == 8 * on_device and st.get("pictures_output") == 8 * SY.MD5[name]["pictures"], st
    SY.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("seed", range(6))
def test_bridge_survives_damaged_streams_emulated(tmp_path, emu, seed):
    """bit errors in slice data: whatever the reference decoder makes of the stream (errors, partial pictures, concealment),
    the bridge neither crashes nor hangs — it finishes what it has and hands the decoder back to the C path"""
    import random
    import struct
    import subprocess
    buf = open(SY.samples("420_8_qcif"), "rb").read()
    p = 4 + struct.unpack_from("<I", buf, 0)[0]
    n = struct.unpack_from("<I", buf, p)[0]
    p += 4
    units = []
    for _ in range(n):
        ln = struct.unpack_from("<I", buf, p)[0]
        units.append(bytearray(buf[p + 4:p + 4 + ln]))
        p += 4 + ln
    r = random.Random(seed)
    for _ in range(r.randint(1, 6)):
        u = units[r.randint(1, n - 1)]
        u[r.randint(40, len(u) - 1)] ^= 1 << r.randint(0, 7)
    src = tmp_path / "d.samples"
    with open(src, "wb") as f:
        f.write(struct.pack("<II", 0, n))
        for u in units:
            f.write(struct.pack("<I", len(u)) + bytes(u))
    for lazy in (False, True):
        env = dict(os.environ)
        env.pop("MI355_BRIDGE_LAZY", None)
        if lazy:
            env["MI355_BRIDGE_LAZY"] = "1"
        res = subprocess.run([SY.exe("h264_bridge_emu"), str(src), str(tmp_path / "o.yuv"), "1", "1"], capture_output=True, text=True, env=env, timeout=600)
        assert res.returncode == 0, (res.returncode, res.stderr[-500:])


@needs_harness
@pytest.mark.parametrize("lazy", (False, True))
def test_bridge_leaves_the_path_in_the_middle_of_a_picture_emulated(tmp_path, emu, lazy):
    """a picture with more slices than the bridge's tables hold (MI355_BRIDGE_MAX_SLICES=8 on the 30-slice stream): the bridge
    gives back what is in flight and its buffers, the decoder finishes the stream on the C path — same exit code and number of
    pictures as the plain run, no hang, and the pictures before the one that overflowed are the reference's"""
    import subprocess
    name = "420_8_slices30"
    env = dict(os.environ, MI355_BRIDGE_MAX_SLICES="8")
    env.pop("MI355_BRIDGE_LAZY", None)
    if lazy:
        env["MI355_BRIDGE_LAZY"] = "1"
    out, ref = tmp_path / "o.yuv", tmp_path / "r.yuv"
    res = subprocess.run([SY.exe("h264_bridge_emu"), SY.samples(name), str(out), "1", "1"], capture_output=True, text=True, env=env, timeout=600)
    plain = subprocess.run([SY.exe("h264_bridge_emu"), SY.samples(name), str(ref), "1", "1"], capture_output=True, text=True,
                           env=dict(os.environ, MI355_BRIDGE_PLAIN="1"), timeout=600)
    assert res.returncode == plain.returncode == 0, (res.returncode, res.stderr[-500:])
    assert "more slices or reference pictures than the batched path holds" in res.stderr
    st = json.loads(res.stdout.strip().splitlines()[-1])
    assert st["bridges_active"] == 0 and st["pictures_output"] == SY.MD5[name]["pictures"], st
    assert os.path.getsize(out) == os.path.getsize(ref)


def _truncate_samples(src, dst, keep):
    """the first `keep` packets of a .samples file (u32 extradata size, extradata, u32 count, count x (u32 size, bytes))"""
    import struct
    raw = open(src, "rb").read()
    el, = struct.unpack_from("<I", raw, 0)
    o = 4 + el
    n, = struct.unpack_from("<I", raw, o)
    o += 4
    head, body, k = raw[:4 + el], b"", min(keep, n)
    for _ in range(k):
        ln, = struct.unpack_from("<I", raw, o)
        body += raw[o:o + 4 + ln]
        o += 4 + ln
    open(dst, "wb").write(head + struct.pack("<I", k) + body)
    return k


@needs_harness
@pytest.mark.parametrize("name", ("444_8_paff", "420_8_paff"))
def test_bridge_lazy_field_pairs_at_the_end_of_a_stream_emulated(tmp_path, emu, name):
    """MI355_BRIDGE_LAZY with PAFF: the two fields of a frame can be in flight in either order of staging sets when the stream
    ends (or is flushed); each brings back its own lines only.  Every truncation of the stream to 2..7 packets — ends in a field
    pair after an odd and after an even number of pictures — must give what the reference's C path gives."""
    import hashlib
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    for keep in range(2, 8):
        src = tmp_path / ("t%d.samples" % keep)
        _truncate_samples(SY.samples(name), src, keep)
        md5 = {}
        for mode, envs in (("plain", {"MI355_BRIDGE_PLAIN": "1"}), ("default", {}), ("lazy", {"MI355_BRIDGE_LAZY": "1"}),
                           ("lazy_direct", {"MI355_BRIDGE_LAZY": "1", "MI355_BRIDGE_DIRECT": "1"})):
            env = dict(os.environ)
            for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_PLAIN"):
                env.pop(k, None)
            env.update(envs)
            out = tmp_path / ("o_%s_%d.yuv" % (mode, keep))
            r = subprocess.run([SY.exe("h264_bridge_emu"), str(src), str(out), "1", "1"], capture_output=True, text=True, env=env, timeout=1800)
            assert r.returncode == 0, r.stderr[-2000:]
            md5[mode] = hashlib.md5(open(out, "rb").read()).hexdigest()
        assert len(set(md5.values())) == 1, (name, keep, md5)


@needs_harness
@pytest.mark.parametrize("name", [n for n in SY.CLASSIC if not n.startswith("444")])
def test_bridge_through_the_session_facade_emulated(tmp_path, emu, name):
    """MI355_BRIDGE_SESSION: the reference decoder hands every picture to mi355_h264_start_frame / decode_slice / end_frame
    (the AVHWAccel-shaped façade) and takes it back with get_frame — frame and field pictures, slices, B pictures, weights"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_emu", name, out, session=True)
    assert st.get("pictures_on_device") == SY.ON_DEVICE.get(name, SY.MD5[name]["pictures"]) and st.get("launch_sets") == 0, st
    SY.check_md5(out, name)


@needs_harness
def test_bridge_decodes_1080p_high10_stream_emulated(tmp_path, emu):
    """tests/golden/h264_synth_1080p_high10.samples (tools/make_1080p_stream.py: 120 x 68 macroblocks, I / P / B with implicit weights, four slices,
    8x8 transform, three references, 10-bit samples): every picture through the second kernel set on the emulator = the reference decoder's own output"""
    import hashlib
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(SY.ROOT, "oracle"), "_ref/h264_bridge_emu"], check=True)
    src = os.path.join(SY.GOLD, "h264_synth_1080p_high10.samples")
    md5 = {}
    for mode, env in (("plain", {"MI355_BRIDGE_PLAIN": "1"}), ("bridge", {})):
        e = dict(os.environ)
        e.pop("MI355_BRIDGE_PLAIN", None)
        e.update(env)
        out = tmp_path / (mode + ".yuv")
        r = subprocess.run([SY.exe("h264_bridge_emu"), src, str(out), "1", "1"], capture_output=True, text=True, env=e, timeout=1800)
        assert r.returncode == 0, r.stderr[-2000:]
        st = json.loads(r.stdout.strip().splitlines()[-1])
        assert st["pictures_output"] == 10 and st["pictures_on_device"] == (10 if mode == "bridge" else 0), st
        md5[mode] = hashlib.md5(open(out, "rb").read()).hexdigest()
    assert md5["plain"] == md5["bridge"]
