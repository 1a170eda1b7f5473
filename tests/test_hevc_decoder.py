"""CPU: the REFERENCE's own HEVC decoder with its three DSP tables (hevcdec.c:443-445: HEVCDSPContext, HEVCPredContext,
VideoDSPContext) overridden through the linker by this project's ff_*_init_mi355x hooks, running the SIMT-emulated build
of the product sources, decodes the generated streams (hevc_streams.py) to what it decodes with its own tables."""
import os
import subprocess

import pytest

import hevc_streams as HS

needs_harness = pytest.mark.skipif(not os.path.isdir("/root/reference/libavcodec"), reason="needs the reference decoder objects (/root/reference)")


def test_streams_cover_both_depths_and_inter_pictures():
    assert {HS.MD5[n]["pix_fmt"] for n in HS.ALL} == {"yuv420p", "yuv420p9le", "yuv420p10le"}
    assert sum(n.startswith("pb_") for n in HS.ALL) >= 6 and sum(n.startswith("i_") for n in HS.ALL) >= 12
    for n in HS.ALL:
        assert os.path.getsize(HS.samples(n)) > 1000


@needs_harness
def test_writer_is_deterministic_and_the_reference_accepts_its_streams(tmp_path):
    """the committed streams are what the writer writes today (tables re-read from the reference's sources)"""
    import hashlib
    import sys
    sys.path.insert(0, HS.GOLD)
    import make_hevc_streams as M
    for name in ("i_8bit", "pb_10bit_weighted"):
        pkts = M.Hevc(name, **M.STREAMS[name]).build()
        assert hashlib.md5(b"".join(pkts)).hexdigest() == HS.MD5[name]["stream_md5"]


@needs_harness
@pytest.mark.parametrize("name", HS.EMU)
def test_reference_hevc_decoder_with_tier1_hooks_emulated(tmp_path, emu, name):
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_tier1_emu"], check=True)
    out = tmp_path / "plain.yuv"
    assert HS.run_tier1("hevc_tier1_emu", name, out, plain=True)[0] == 0
    HS.check_md5(out, name)                                      # the committed md5 is the reference's
    out = tmp_path / "hooked.yuv"
    assert HS.run_tier1("hevc_tier1_emu", name, out)[0] >= 150   # the hooks filled the tables (169 entries at this revision)
    HS.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name", HS.EMU)
def test_reference_hevc_decoder_with_picture_level_filters_emulated(tmp_path, emu, name):
    """contrib/libav/mi355_hevc_lf_bridge.c: boundary strengths, deblocking and SAO of every picture in one device pass each, fed with the arrays
    the reference's slice decoder leaves behind — the whole sequence (P / B pictures predict from the filtered pictures)
    equals the reference's; with the DSP tables hooked as well, and with the reference's own tables"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_lf_emu"], check=True)
    for plain in (False, True):
        out = tmp_path / "lf.yuv"
        hooks, pictures = HS.run_tier1("hevc_lf_emu", name, out, plain=plain)
        assert pictures == HS.MD5[name]["pictures"] and (hooks == 0) == plain
        HS.check_md5(out, name)
    out = tmp_path / "lf_plain.yuv"
    assert HS.run_tier1("hevc_lf_emu", name, out, plain=True, lf_plain=True) == (0, 0)
    HS.check_md5(out, name)


@needs_harness
@pytest.mark.parametrize("name", HS.EMU)
def test_batched_intra_wrapper_inside_the_reference_decoder_emulated(tmp_path, emu, name):
    """HEVCPredContext.intra_pred[] replaced by mi355_hevc_intra_pred_blocks_dev(), one block per call, on the decoder's own
    state (picture so far, lc->na, tab_mvf with constrained intra prediction, min_tb_addr_zs): every intra block of every
    stream — a pin of the batched wrapper, not a binding (oracle/ref_hevc_tier1_main.c)"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_tier1_emu"], check=True)
    out = tmp_path / "intra.yuv"
    assert HS.run_tier1("hevc_tier1_emu", name, out, intra_device=True) >= 100
    HS.check_md5(out, name)
