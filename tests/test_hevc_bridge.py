"""CPU (emulated kernels), where /root/reference exists: the HEVC Tier-2 bridge — the reference's own HEVC decoder with every
prediction block, transform unit, intra block and PCM block RECORDED through its three pointer tables and run on the (emulated)
device level by level, the in-loop filters of the picture on the same device picture, references in device memory
(contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c; VERDICT r2 "what's missing" 1).  Every generated stream — I / P / B,
all transform sizes and scans, Intra NxN, PCM, transform skip, transquant bypass with the filters on, SAO with merging, slices,
AMP, merge / AMVP, explicit weights, constrained intra prediction, 8 and 10 bit — must come out as from the unmodified decoder."""
import os
import subprocess

import pytest

import hevc_streams as HS

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/libavcodec"), reason="needs the reference decoder objects (/root/reference)")


@pytest.mark.parametrize("name", HS.EMU)
def test_hevc_bridge_decodes_generated_streams_emulated(tmp_path, emu, name):
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_emu", name, out)
    n = HS.MD5[name]["pictures"]
    assert st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n and st["pictures_filtered_on_device"] == n, st
    assert st["reference_uploads"] == 0, st                               # references never crossed the link
    HS.check_md5(out, name)


def test_hevc_bridge_plain_run_is_the_reference_path(tmp_path, emu):
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    name = HS.ALL[0]
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_emu", name, out, plain=True)
    assert st["pictures_reconstructed_on_device"] == 0 and st["pictures_filtered_on_device"] == 0
    HS.check_md5(out, name)


@pytest.mark.parametrize("name", ["pb_8bit", "pb_10bit_weighted", "pb_480p_ctb64"])
def test_hevc_bridge_random_access_pictures_on_the_host_emulated(tmp_path, emu, name):
    """MI355_HEVC_BRIDGE_IRAP_ON_HOST=1 (a scheduling policy): the stream's first picture — all intra, a long chain of dependent blocks —
    is reconstructed by the reference's functions, filtered on the device like every picture, and uploaded ONCE when the next picture
    predicts from it; everything else as without the policy, the output the reference's"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_emu", name, out, irap_on_host=True)
    n = HS.MD5[name]["pictures"]
    assert st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n - 1 and st["pictures_filtered_on_device"] == n, st
    assert st["reference_uploads"] == 1, st
    HS.check_md5(out, name)


@pytest.mark.parametrize("name", ["i_ctb64", "pb_tiles_dep", "i_pcm_lf_off_10bit"])
def test_hevc_bridge_intra_blocks_with_their_residual_in_one_launch_emulated(tmp_path, emu, name):
    """the default: an intra block's transform unit rides in the launch of its prediction (mi355_hevc_intra_recon_blocks_dev) — fewer
    dependency levels and launches than MI355_HEVC_BRIDGE_SPLIT_INTRA=1 (the two-launch form), the same pictures"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    fused = HS.run_bridge("hevc_bridge_emu", name, tmp_path / "f.yuv")
    HS.check_md5(tmp_path / "f.yuv", name)
    split = HS.run_bridge("hevc_bridge_emu", name, tmp_path / "s.yuv", split_intra=True)
    HS.check_md5(tmp_path / "s.yuv", name)
    assert fused["dependency_levels"] < split["dependency_levels"] and fused["reconstruction_launches"] < split["reconstruction_launches"], (fused, split)


@pytest.mark.parametrize("name", ["i_ctb64", "pb_tiles_dep", "pb_10bit_weighted", "pb_480p_ctb64"])
def test_hevc_bridge_all_levels_in_one_launch_emulated(tmp_path, emu, name):
    """MI355_HEVC_BRIDGE_ONE_LAUNCH=1: every dependency level of a picture's launch set through mi355_hevc_recon_levels_dev — one reconstruction launch per set, the
    same pictures (on the emulator workgroups run in ticket order: what the waits order is the device test's business)"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    st = HS.run_bridge("hevc_bridge_emu", name, tmp_path / "o.yuv", one_launch=True)
    HS.check_md5(tmp_path / "o.yuv", name)
    assert st["pictures_reconstructed_on_device"] == HS.MD5[name]["pictures"] and st["reconstruction_launches"] <= st["launch_sets"] < st["dependency_levels"], st


def test_small_pictures_stay_on_the_host_by_default(emu, tmp_path):
    """the bridges' size policy (MI355_HEVC_BRIDGE_MIN_PIXELS, default 1.5 M luma samples): a 96x64 stream decoded WITHOUT the tests' override is left to
    the reference's C path — same pictures, nothing reconstructed on the device — because one decoder's launch set per dependency level costs more than the C
    functions on small pictures (below 1080p the bridge was a slow-down of up to 27x)"""
    import subprocess
    import json
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    name = "pb_8bit"
    out = tmp_path / "o.yuv"
    env = dict(os.environ)
    for k in ("MI355_HEVC_RECON_PLAIN", "MI355_HEVC_LF_PLAIN", "MI355_HEVC_BRIDGE_MIN_PIXELS"):
        env.pop(k, None)
    r = subprocess.run([os.path.join(HS.ROOT, "oracle", "_ref", "hevc_bridge_emu"), HS.samples(name), str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stderr.strip() == "", r.stderr[-1000:]
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["pictures_output"] > 0 and st["pictures_reconstructed_on_device"] == 0
    HS.check_md5(out, name)


@pytest.mark.parametrize("name,threads", [(n, t) for n in HS.EMU[::4] for t in (3,)] + [(HS.EMU[1], 7)])
def test_hevc_bridge_many_decoders_share_launches_emulated(tmp_path, emu, name, threads):
    """several decoders in one process (one per thread, the same stream twice over): a picture's launches are issued together with those of every
    other decoder's picture that is waiting at that moment (contrib/libav/mi355_hevc_bridge.c commit_launches: level l of the whole batch is one
    launch per job kind) — every decoder outputs what the unmodified decoder does, every picture is reconstructed and filtered on the device, and
    the process needs fewer launches than the decoders would alone (MI355_HEVC_BRIDGE_SOLO=1: the same run, every picture its own launches)"""
    subprocess.run(["make", "-s", "-C", os.path.join(HS.ROOT, "oracle"), "_ref/hevc_bridge_emu"], check=True)
    out = tmp_path / "o.yuv"
    st = HS.run_bridge("hevc_bridge_emu", name, out, threads=threads, loops=2)
    n = HS.MD5[name]["pictures"] * threads * 2
    assert st["threads"] == threads and st["outputs_identical"] is True, st
    assert st["pictures_output"] == n and st["pictures_reconstructed_on_device"] == n and st["pictures_filtered_on_device"] == n, st
    HS.check_md5(out, name)
    solo = HS.run_bridge("hevc_bridge_emu", name, tmp_path / "s.yuv", threads=threads, loops=2, solo=True)
    assert solo["outputs_identical"] is True and solo["pictures_per_launch_set"] == 1.0, solo
    HS.check_md5(tmp_path / "s.yuv", name)
    assert st["reconstruction_launches"] <= solo["reconstruction_launches"], (st, solo)
