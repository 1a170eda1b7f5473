"""Generated H.264 streams (tests/golden/make_h264_streams.py: a CAVLC bitstream writer with random syntax elements) for
what the offline clips lack: several slices per picture, the loop filter off at slice edges / altogether, I_PCM, explicit
weights in 4:2:0, four references, every partition shape, far vectors — and the profiles no clip has: High 4:2:2,
High 10, High 4:2:2 at 10 bit, 9 bit, High 4:4:4 Predictive with slices / I_PCM (8 and 10 bit), transform bypass (lossless) in 4:2:0, 4:4:4 and 10-bit 4:2:2; cropped pictures; constrained intra prediction with I and P
slices in one picture; sequences that change picture size / format; interlaced-capable sequences with field pictures (PAFF: Tier 2 decodes them too) and with macroblock pairs coded as frame / field macroblocks (MBAFF).  tests/golden/h264_synth_ref_md5.json = md5 of what the reference's own decoder
(tables untouched) outputs for each; tests/golden/h264_stream_synth_*.npz = the Tier-2 records the reference decoder's run
exported for seven of the 8-bit 4:2:0 streams (four of them with B pictures: implicit weights, explicit weights, plain average; one with the 8x8 transform and Intra 8x8) (oracle/ref_h264_export.c), as for realshort.mp4."""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MD5 = json.load(open(os.path.join(GOLD, "h264_synth_ref_md5.json")))
ALL = sorted(MD5)
# the two places where the reference's decoder is not consistent with itself: the bridge hands such pictures to the C path
QUIRKS = ["420_8_2wide_b", "420_8_paff_idc2_intra"]
CLASSIC = [n for n in ALL if n.startswith(("420_8_", "444_8")) and "lossless" not in n and "mbaff" not in n and n not in QUIRKS]            # 8-bit 4:2:0 and 4:4:4: the first kernel set
# High 10 (9 / 10 bit), High 4:2:2 (8 / 10 bit), 4:4:4 at 10 bit, and transform bypass at any of these and at 8-bit 4:2:0 / 4:4:4: the second kernel set
# (mi355_h264_decode_frames_wide_dev); MBAFF frames (macroblock pairs) at every format too
WIDE = [n for n in ALL if n.startswith(("420_10", "420_9", "422_", "444_10")) or "lossless" in n or "mbaff" in n]
BRIDGE = CLASSIC + WIDE                                                                                                   # what Tier 2 decodes
OUTSIDE = [n for n in ALL if n not in BRIDGE and n not in ("mixed_formats", "paff_and_frames") and n not in QUIRKS]                           # nothing any more
# field pictures: the bridge counts pictures (a frame coded as two fields is two), the md5 file counts output frames
ON_DEVICE = {"420_8_paff": 13, "420_8_paff_b": 19, "420_8_paff_t8x8": 14, "444_8_paff": 8, "422_10_paff": 10}
EXPORTED = ["420_8_slices", "420_8_qcif", "420_8_nofilter", "420_8_b_implicit", "420_8_b_explicit", "420_8_b_average", "420_8_t8x8", "420_8_cip_mixed", "420_8_reorder_b"]


def samples(name):
    return os.path.join(GOLD, "h264_synth_%s.samples" % name)


def npz(name):
    return os.path.join(GOLD, "h264_stream_synth_%s.npz" % name)


def exe(which):
    return os.path.join(ROOT, "oracle", "_ref", which)


def run_tier1(which, name, out, plain=False):
    env = dict(os.environ)
    env.pop("MI355_TIER1_PLAIN", None)
    if plain:
        env["MI355_TIER1_PLAIN"] = "1"
    r = subprocess.run([exe(which), samples(name), str(out)], capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if l.strip()]
    assert len(lines) == 1 and "%d pictures" % MD5[name]["pictures"] in lines[0], r.stderr[-2000:]      # no decoder complaint
    return lines[0]


def run_bridge(which, name, out, threads=1, lazy=False, direct=False, loops=1, session=False, no_wide=False, keep_field_idc2=True):
    env = dict(os.environ)
    # field pictures with disable_deblocking_filter_idc 2 stay on the device in every test but the one of the hand-over (the fixtures' streams are ones on which the
    # reference's inconsistency does not show: identical output either way)
    env.pop("MI355_BRIDGE_KEEP_FIELD_IDC2", None)
    if keep_field_idc2:
        env["MI355_BRIDGE_KEEP_FIELD_IDC2"] = "1"
    for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_PLAIN", "MI355_BRIDGE_SESSION", "MI355_BRIDGE_LINEAR", "MI355_BRIDGE_NO_WIDE"):
        env.pop(k, None)
    if no_wide:
        env["MI355_BRIDGE_NO_WIDE"] = "1"         # the bridge without the second kernel set: High 10 / High 4:2:2 stay on the C path
    if session:
        env["MI355_BRIDGE_SESSION"] = "1"         # pictures through the whole-frame session façade (mi355_h264_session.h)
    if lazy:
        env["MI355_BRIDGE_LAZY"] = "1"
    if direct:
        env["MI355_BRIDGE_DIRECT"] = "1"
    r = subprocess.run([exe(which), samples(name), str(out), str(threads), str(loops)], capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    stats = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return stats[-1] if stats else {}


def check_md5(path, name):
    raw = open(path, "rb").read()
    assert len(raw) == MD5[name]["bytes"], (len(raw), MD5[name])
    assert hashlib.md5(raw).hexdigest() == MD5[name]["md5"], "%s: pictures differ from the reference decoder's" % name
