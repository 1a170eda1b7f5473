"""Generated HEVC streams (tests/golden/make_hevc_streams.py: a CABAC bitstream writer with random syntax elements; the
reference tree holds no HEVC sample): I pictures with every transform size 4..32, the three coefficient scans, Intra NxN,
SAO band / edge with merging, deblocking offsets / off, cu_qp_delta, transform skip, transquant bypass, default scaling
lists, several slices, tiles (uniform and explicit grids, filtering across their edges on and off, slices of whole tiles and
tiles of whole slices), wavefronts, dependent slice segments, CTB sizes 16 / 32 / 64, 8 and 10 bit — and P / B pictures: skip, merge, AMVP with random vector
differences, all partition shapes (AMP too), one or two lists, explicit weights, constrained intra prediction, groups of
pictures decoded out of output order with references on both sides.
tests/golden/hevc_streams.json = md5 of what the reference's own decoder (tables untouched) outputs for each, and the
number of coding tree units / slices written (the decoder must see exactly those: the streams are open loop)."""
import hashlib
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MD5 = json.load(open(os.path.join(GOLD, "hevc_streams.json")))
ALL = sorted(MD5)
EMU = [n for n in ALL if not n.startswith("pb_1080p")]
# pictures whose slices use different reference lists: the filter bridge takes the decoder's own boundary strengths for them (the calls of
# ff_hevc_deblocking_boundary_strengths it had put aside are run when the second list set shows up)
MIXED_LISTS = {"pb_slices_own_lists"}        # the SIMT emulator takes half a minute per pass over the 1080p stream: GPU tests only


def samples(name):
    return os.path.join(GOLD, "hevc_synth_%s.samples" % name)


def run_tier1(which, name, out, plain=False, lf_plain=False, intra_device=False):
    """-> (table entries the hooks replaced, pictures filtered by the picture-level pass); asserts that those pictures took
    their boundary strengths from the device pass too (one reference list set per picture in all streams), that the decoder raised no
    complaint and saw exactly the coding tree units and slice ends the writer wrote (the arithmetic decoding stayed in step)"""
    env = dict(os.environ)
    env.pop("MI355_TIER1_PLAIN", None)
    env.pop("MI355_HEVC_LF_PLAIN", None)
    if plain:
        env["MI355_TIER1_PLAIN"] = "1"
    if lf_plain:
        env["MI355_HEVC_LF_PLAIN"] = "1"
    env.pop("MI355_HEVC_INTRA_DEVICE", None)
    env["MI355_HEVC_BRIDGE_MIN_PIXELS"] = "0"           # the generated streams are small: the size policy of the bridges must not send them to the C path
    if intra_device:
        env["MI355_HEVC_INTRA_DEVICE"] = "1"
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", which), samples(name), str(out)], capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if l.strip()]
    assert len(lines) == 1 and "%d pictures" % MD5[name]["pictures"] in lines[0], r.stderr[-2000:]
    assert "%d coding tree units in %d slices" % (MD5[name]["ctus"], MD5[name]["slices"]) in lines[0], lines[0]
    if intra_device:
        return int(re.search(r"(\d+) intra blocks predicted by the batched wrapper", lines[0]).group(1))
    lf = re.search(r"(\d+) pictures deblocked per picture \((\d+) with strengths from the device\)", lines[0])
    if name in MIXED_LISTS and int(lf.group(1)):
        assert 0 < int(lf.group(2)) < int(lf.group(1)), lines[0]
    else:
        assert lf.group(1) == lf.group(2), lines[0]
    return int(re.search(r"\((\d+) entries replaced\)", lines[0]).group(1)), int(lf.group(1))


def check_md5(path, name):
    raw = open(path, "rb").read()
    assert len(raw) == MD5[name]["bytes"], (len(raw), MD5[name])
    assert hashlib.md5(raw).hexdigest() == MD5[name]["md5"], "%s: pictures differ from the reference decoder's" % name


def run_bridge(which, name, out, plain=False, irap_on_host=False, split_intra=False, threads=1, loops=1, solo=False, one_launch=False):
    """the reference's HEVC decoder with contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c (oracle/_ref/hevc_bridge_{emu,gpu}): -> its
    JSON counters; plain = the comparison run (everything forwarded to the reference's own functions)"""
    env = dict(os.environ)
    for k in ("MI355_HEVC_RECON_PLAIN", "MI355_HEVC_LF_PLAIN", "MI355_HEVC_BS_HOST", "MI355_HEVC_BRIDGE_IRAP_ON_HOST", "MI355_HEVC_BRIDGE_SPLIT_INTRA", "MI355_HEVC_BRIDGE_SOLO", "MI355_HEVC_BRIDGE_ONE_LAUNCH"):
        env.pop(k, None)
    env["MI355_HEVC_BRIDGE_MIN_PIXELS"] = "0"           # small streams on the device all the same (default: pictures below 1.5 M samples stay on the host)
    if plain:
        env["MI355_HEVC_RECON_PLAIN"] = env["MI355_HEVC_LF_PLAIN"] = "1"
    if irap_on_host:
        env["MI355_HEVC_BRIDGE_IRAP_ON_HOST"] = "1"
    if split_intra:
        env["MI355_HEVC_BRIDGE_SPLIT_INTRA"] = "1"      # an intra block's prediction and its residual as two launches
    if one_launch:
        env["MI355_HEVC_BRIDGE_ONE_LAUNCH"] = "1"       # every dependency level of a launch set in one launch (mi355_hevc_recon_levels_dev)
    if solo:
        env["MI355_HEVC_BRIDGE_SOLO"] = "1"             # every picture issues its own launches (no sharing between the decoders of the process)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", which), samples(name), str(out), str(loops), str(threads)], capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.strip() == "", r.stderr[-2000:]                       # no decoder complaint, no bridge failure
    return json.loads(r.stdout.strip().splitlines()[-1])
