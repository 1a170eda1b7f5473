"""GPU (HEVC): the REFERENCE's own HEVC decoder, its three DSP tables overridden through the linker
by this project's ff_*_init_mi355x hooks and bound to the real HIP library, decodes the generated streams (hevc_streams.py)
on the MI355X to what the unmodified reference decoder produced (tests/golden/hevc_streams.json).
oracle/_ref/hevc_tier1_gpu is built where /root/reference exists (oracle/Makefile, __graft_entry__.build()) and travels to
the GPU box with the tree."""
import os

import pytest

import hevc_streams as HS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", HS.ALL)
def test_reference_hevc_decoder_on_gpu(tmp_path, mi355, name):
    if not os.path.exists(os.path.join(HS.ROOT, "oracle", "_ref", "hevc_tier1_gpu")):
        pytest.fail("oracle/_ref/hevc_tier1_gpu missing: run __graft_entry__.build() where /root/reference exists")
    out = tmp_path / "hooked.yuv"
    assert HS.run_tier1("hevc_tier1_gpu", name, out)[0] >= 150
    HS.check_md5(out, name)


@pytest.mark.parametrize("name", HS.ALL)
def test_reference_hevc_decoder_with_picture_level_filters_on_gpu(tmp_path, mi355, name):
    """contrib/libav/mi355_hevc_lf_bridge.c: the reference's tables untouched, deblocking + SAO per picture on the MI355X"""
    if not os.path.exists(os.path.join(HS.ROOT, "oracle", "_ref", "hevc_lf_gpu")):
        pytest.fail("oracle/_ref/hevc_lf_gpu missing: run __graft_entry__.build() where /root/reference exists")
    out = tmp_path / "lf.yuv"
    hooks, pictures = HS.run_tier1("hevc_lf_gpu", name, out, plain=True)
    assert hooks == 0 and pictures == HS.MD5[name]["pictures"]
    HS.check_md5(out, name)


@pytest.mark.parametrize("name", ["i_8bit", "i_10bit", "pb_ctb16_slices_cip"])
def test_batched_intra_wrapper_inside_the_reference_decoder_on_gpu(tmp_path, mi355, name):
    """intra_pred[] replaced by mi355_hevc_intra_pred_blocks_dev(), one block per call, on the decoder's own state"""
    out = tmp_path / "intra.yuv"
    assert HS.run_tier1("hevc_tier1_gpu", name, out, intra_device=True) >= 100
    HS.check_md5(out, name)
