"""swscale oracle pins (CPU): the inner loops and whole pictures against golden vectors produced by
the reference's own libswscale objects, the yuv->rgb LUT construction against the LUTs the reference
built, and — where /root/reference exists — against the objects themselves over more seeds."""
import ctypes as C
import hashlib
import json
import os

import pytest

import cases_sws
import sws_support as S

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sws_ref_sha1.json")))


def sha(b):
    return hashlib.sha1(b).hexdigest()[:20]


@pytest.fixture(scope="module")
def sws_ref():
    r = S.reference()
    if r is None:
        pytest.skip("/root/reference not present")
    return r


def test_oracle_functions_match_reference_golden(oracle):
    luts = S.load_context("down2_128x96").desc.luts
    assert sha(bytes(luts)) == GOLD["luts"]
    got = cases_sws.run_functions(cases_sws.Funcs(oracle.lib, "oracle_sws_"), luts, GOLD["seed"])
    for name, want in GOLD["functions"].items():
        assert sha(got[name]) == want, name
    assert any(k.startswith("c24/") for k in got)


def test_oracle_lut_construction_matches_reference(oracle):
    """ff_yuv2rgb_c_init_tables with the default ITU-601 coefficients, limited range
    (what sws_getContext sets up: yuv2rgb.c:49-58 row SWS_CS_DEFAULT, utils.c:807)"""
    lut = S.Luts()
    inv = (C.c_int * 4)(104597, 132201, 25675, 53279)
    oracle.lib.oracle_sws_init_luts(C.byref(lut), inv, 0, 0, 1 << 16, 1 << 16)
    assert sha(bytes(lut)) == GOLD["luts"]


def test_oracle_reproduces_fate_pixfmt_rgb24(oracle):
    """fate-pixfmt-rgb24 (tests/fate/pixfmt.mak, fate-run.sh:236-246): vsynth1 frame 0, yuv420p ->
    rgb24 with bicubic+accurate_rnd+bitexact.  The rgb24 stage is this path; its sha1 was recorded
    together with the proof that the reference's rgb24 -> yuv444p second stage turns it into the md5
    stored in tests/ref/pixfmt/rgb24 (c6e0f9b5...).  Where the reference is present the whole chain
    is redone."""
    ctx = S.load_context("cif_generic")
    rgb = S.oracle_backend(oracle).scale(ctx, S.fate_frame())
    assert sha(rgb.tobytes()) == GOLD["pictures"]["fate_pixfmt_rgb24_stage1"]
    assert GOLD["fate_pixfmt_rgb24_md5"] == "c6e0f9b5817f484b175c1ec4ffb4e9c9"
    ref = S.reference()
    if ref is not None:
        assert S.fate_chain_md5(ref, rgb) == open("/root/reference/tests/ref/pixfmt/rgb24").read().split()[0]


@pytest.mark.parametrize("name", list(S.CONFIGS))
def test_oracle_picture_matches_reference_golden(oracle, name):
    ctx = S.load_context(name)
    got = S.oracle_backend(oracle).scale(ctx, S.picture(name), dst_pad=8)
    assert sha(got.tobytes()) == GOLD["pictures"][name]


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_matches_reference_objects(oracle, sws_ref, seed):
    rf = cases_sws.RefFuncs(sws_ref)
    want = cases_sws.run_functions(rf, rf.luts, seed)
    got = cases_sws.run_functions(cases_sws.Funcs(oracle.lib, "oracle_sws_"), rf.luts, seed)
    assert [k for k in want if want[k] != got[k]] == []
    for name in S.SMALL:
        ctx = sws_ref.context(name)
        planes = S.picture(name, seed, stride_pad=3)
        assert (sws_ref.scale(name, planes) == S.oracle_backend(oracle).scale(ctx, planes)).all(), name
