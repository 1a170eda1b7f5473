"""GPU: the HEVC Tier-2 bridge (contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c) inside the reference's own HEVC decoder
bound to the real library (oracle/_ref/hevc_bridge_gpu, built HERE by __graft_entry__.build(); /root/reference is not read on the
GPU box): every generated stream reconstructed and filtered on the MI355X, references in HBM, output identical to the unmodified
decoder's (tests/golden/hevc_streams.json)."""
import os

import pytest

import hevc_streams as HS

pytestmark = pytest.mark.gpu
EXE = os.path.join(HS.ROOT, "oracle", "_ref", "hevc_bridge_gpu")


@pytest.mark.parametrize("name", HS.ALL)
def test_hevc_bridge_decodes_generated_streams_gpu(tmp_path, mi355, name):
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/hevc_bridge_gpu missing: __graft_entry__.build() makes it where /root/reference exists")
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_gpu", name, out)
    n = HS.MD5[name]["pictures"]
    assert st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n and st["pictures_filtered_on_device"] == n, st
    assert st["reference_uploads"] == 0, st
    HS.check_md5(out, name)


def test_hevc_bridge_plain_run_gpu(tmp_path, mi355):
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_gpu", "pb_8bit", out, plain=True)
    assert st["pictures_reconstructed_on_device"] == 0 and st["pictures_filtered_on_device"] == 0
    HS.check_md5(out, "pb_8bit")


@pytest.mark.parametrize("name", ["pb_8bit", "pb_480p_ctb64", "pb_1080p_few_intra"])
def test_hevc_bridge_random_access_pictures_on_the_host_gpu(tmp_path, mi355, name):
    """MI355_HEVC_BRIDGE_IRAP_ON_HOST=1: the all-intra first picture stays with the reference's functions, is filtered on the device and
    uploaded once when the next picture predicts from it; output identical"""
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_gpu", name, out, irap_on_host=True)
    n = HS.MD5[name]["pictures"]
    assert st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n - 1 and st["pictures_filtered_on_device"] == n, st
    assert st["reference_uploads"] == 1, st
    HS.check_md5(out, name)


@pytest.mark.parametrize("name", ["i_ctb64", "pb_tiles_dep", "pb_1080p_few_intra"])
def test_hevc_bridge_two_launch_form_of_intra_blocks_gpu(tmp_path, mi355, name):
    """MI355_HEVC_BRIDGE_SPLIT_INTRA=1: prediction and residual of an intra block as two launches (the default fuses them,
    mi355_hevc_intra_recon_blocks_dev: the test above) — the same pictures from more dependency levels"""
    out = tmp_path / "o.yuv"
    split = HS.run_bridge("hevc_bridge_gpu", name, out, split_intra=True)
    HS.check_md5(out, name)
    fused = HS.run_bridge("hevc_bridge_gpu", name, tmp_path / "f.yuv")
    assert fused["dependency_levels"] < split["dependency_levels"], (fused, split)


@pytest.mark.parametrize("name,threads", (("i_ctb64", 1), ("pb_tiles_dep", 1), ("pb_10bit_weighted", 1), ("pb_480p_ctb64", 4), ("pb_1080p_few_intra", 2)))
def test_hevc_bridge_all_levels_in_one_launch_gpu(tmp_path, mi355, name, threads):
    """MI355_HEVC_BRIDGE_ONE_LAUNCH=1: every dependency level of a launch set through mi355_hevc_recon_levels_dev (hundreds to thousands of levels waiting for each
    other inside ONE launch, several sets side by side on their streams when there are several decoders) — the reference decoder's pictures"""
    st = HS.run_bridge("hevc_bridge_gpu", name, tmp_path / "o.yuv", one_launch=True, threads=threads, loops=2)
    HS.check_md5(tmp_path / "o.yuv", name)
    n = HS.MD5[name]["pictures"] * threads * 2
    assert st["outputs_identical"] is True and st["pictures_reconstructed_on_device"] == n and st["reconstruction_launches"] <= st["launch_sets"] < st["dependency_levels"], st


@pytest.mark.parametrize("name,threads", (("pb_8bit", 4), ("i_10bit", 3), ("pb_480p_ctb64", 8)))
def test_hevc_bridge_many_decoders_share_launches_gpu(tmp_path, mi355, name, threads):
    """several decoders in one process (one per thread): the pictures that wait together are launched together (commit_launches) — every
    decoder's output identical to the unmodified decoder's, everything on the device, fewer launches than with MI355_HEVC_BRIDGE_SOLO=1"""
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_gpu", name, out, threads=threads, loops=3)
    n = HS.MD5[name]["pictures"] * threads * 3
    assert st["outputs_identical"] is True and st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n and st["pictures_filtered_on_device"] == n, st
    HS.check_md5(out, name)
    solo = HS.run_bridge("hevc_bridge_gpu", name, tmp_path / "s.yuv", threads=threads, loops=3, solo=True)
    assert solo["outputs_identical"] is True and solo["pictures_per_launch_set"] == 1.0, solo
    assert st["reconstruction_launches"] <= solo["reconstruction_launches"], (st, solo)

