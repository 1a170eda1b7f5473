"""GPU: the generated streams (synth_streams.py) through the kernels, sessions, the Tier-1 hooks and the Tier-2 bridge."""
import os

import numpy as np
import pytest

import h264_frames as HF
import session_cases as SC
import stream_fixture as SF
import synth_streams as SY

pytestmark = pytest.mark.gpu


def _need(which):
    if not os.path.exists(SY.exe(which)):
        pytest.fail("oracle/_ref/%s missing: run __graft_entry__.build() where /root/reference exists" % which)


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_gpu_reproduces_reference_decoder_on_generated_streams(mi355, name):
    pics = SF.load_npz(SY.npz(name))
    d = HF.DeviceFrames(mi355, SF.frameset_all(pics))
    try:
        d.decode()
        got = d.fetch(d.dst)
    finally:
        d.free()
    for f in range(len(pics)):
        for p, key in enumerate(("y", "cb", "cr")):
            assert np.array_equal(got[p][f], pics[f][key]), "picture %d plane %s differs from the reference decoder" % (f, key)


@pytest.mark.parametrize("name", SY.EXPORTED)
def test_session_decodes_generated_streams_in_sequence_gpu(mi355, name):
    SC.run_stream(mi355, SY.npz(name), 0, None, nsurf=8, sync_each=False)


@pytest.mark.parametrize("name", SY.ALL)
def test_reference_decoder_with_tier1_hooks_gpu(tmp_path, mi355, name):
    """High 4:2:2, High 10, 4:2:2 at 10 bit, 9 bit and the 8-bit 4:2:0 streams: hooked = the reference's own output"""
    _need("h264_tier1_gpu")
    out = tmp_path / "hooked.yuv"
    line = SY.run_tier1("h264_tier1_gpu", name, out)
    assert SY.MD5[name]["summary"] in line
    SY.check_md5(out, name)


@pytest.mark.parametrize("lazy,direct,threads", ((False, False, 1), (True, False, 3)))
@pytest.mark.parametrize("name", SY.BRIDGE)
def test_bridge_decodes_generated_streams_gpu(tmp_path, mi355, name, lazy, direct, threads):
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, threads=threads, lazy=lazy, direct=direct)
    assert st.get("pictures_on_device") == threads * SY.ON_DEVICE.get(name, SY.MD5[name]["pictures"]), st
    SY.check_md5(out, name)


@pytest.mark.parametrize("name", [n for n in SY.CLASSIC if not n.startswith("444")])
def test_bridge_through_the_session_facade_gpu(tmp_path, mi355, name):
    """MI355_BRIDGE_SESSION: the reference decoder's pictures through mi355_h264_start_frame / decode_slice / end_frame /
    get_frame (the façade's reference-side caller), two decoder threads = two sessions"""
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, threads=2, session=True)
    assert st.get("pictures_on_device") == 2 * SY.ON_DEVICE.get(name, SY.MD5[name]["pictures"]) and st.get("launch_sets") == 0, st
    SY.check_md5(out, name)


@pytest.mark.parametrize("name,no_wide", (("422_8_b", True), ("420_10_t8x8", True), ("444_10", True), ("420_8_lossless", True), ("444_8_lossless", True), ("422_10_lossless", True),
                                          ("422_10_paff", True), ("420_8_mbaff", True), ("444_8_mbaff", True), ("422_10_mbaff", True)))
def test_bridge_steps_aside_for_streams_outside_tier2_gpu(tmp_path, mi355, name, no_wide):
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, no_wide=no_wide)
    assert st.get("pictures_on_device") == 0 and st.get("pictures_output") == SY.MD5[name]["pictures"], st
    SY.check_md5(out, name)


@pytest.mark.parametrize("name,on_device,frames", (("mixed_formats", 12, 12), ("paff_and_frames", 23, 17)))
@pytest.mark.parametrize("lazy", (False, True))
def test_bridge_follows_sequence_changes_gpu(tmp_path, mi355, lazy, name, on_device, frames):
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, lazy=lazy)
    assert st.get("pictures_on_device") == on_device and st.get("pictures_output") == frames, st
    SY.check_md5(out, name)


@pytest.mark.parametrize("name,on_device", (("420_8_resize", 11), ("mixed_formats", 12)))
def test_bridge_sequence_changes_with_several_decoders_gpu(tmp_path, mi355, name, on_device):
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, threads=6, loops=3)
    assert st.get("pictures_on_device") == 18 * on_device and st.get("pictures_output") == 18 * SY.MD5[name]["pictures"], st
    SY.check_md5(out, name)


@pytest.mark.parametrize("fn", ("h264_synth_1080p.samples", "h264_synth_1080p_high10.samples"))
def test_bridge_decodes_1080p_streams_gpu(tmp_path, mi355, fn):
    """the generated 1080p streams bench.py's real-stream points decode (8 bit: first kernel set, tiled surfaces; High 10: second kernel set), four
    decoder threads: the bridge's pictures = the same binary's with everything left to the reference's C functions"""
    import hashlib
    import json
    import subprocess
    _need("h264_bridge_gpu")
    src = os.path.join(SY.GOLD, fn)
    md5 = {}
    for mode, env in (("plain", {"MI355_BRIDGE_PLAIN": "1"}), ("bridge", {})):
        e = dict(os.environ)
        e.pop("MI355_BRIDGE_PLAIN", None)
        e.update(env)
        out = tmp_path / (mode + ".yuv")
        r = subprocess.run([SY.exe("h264_bridge_gpu"), src, str(out), "4", "1"], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        st = json.loads(r.stdout.strip().splitlines()[-1])
        assert st["pictures_output"] == 40 and st["pictures_on_device"] == (40 if mode == "bridge" else 0), st
        md5[mode] = hashlib.md5(open(out, "rb").read()).hexdigest()
    assert md5["plain"] == md5["bridge"]


@pytest.mark.parametrize("name,on_device", (("420_8_2wide_b", 0), ("420_8_paff_idc2_intra", None)))
def test_bridge_leaves_the_reference_s_inconsistent_cases_to_it_gpu(tmp_path, mi355, name, on_device):
    """as tests/test_synth_streams.py: pictures two macroblocks wide (h264_mb.c:407-409) and field pictures with disable_deblocking_filter_idc 2 (:525-527) are
    handed back to the reference's decoder; the output is the reference's"""
    _need("h264_bridge_gpu")
    out = tmp_path / "o.yuv"
    st = SY.run_bridge("h264_bridge_gpu", name, out, keep_field_idc2=False)
    assert st.get("pictures_output") == SY.MD5[name]["pictures"], st
    if on_device is None:
        assert 0 < st.get("pictures_on_device") < SY.MD5[name]["pictures"], st
    else:
        assert st.get("pictures_on_device") == on_device, st
    SY.check_md5(out, name)
