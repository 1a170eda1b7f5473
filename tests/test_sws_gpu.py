"""GPU: the swscale kernels through the C ABI of include/mi355_sws.h — bit-exact against the oracle,
the golden vectors the reference's own libswscale produced (inner loops, whole pictures up to
SURVEY.md §8d config 5's full sizes) and the first stage of fate-pixfmt-rgb24."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases_sws
import sws_support as S

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sws_ref_sha1.json")))


def sha(b):
    return hashlib.sha1(b).hexdigest()[:20]


def test_gpu_inner_loops_match_golden_and_oracle(mi355, oracle):
    luts = S.load_context("down2_128x96").desc.luts
    got = cases_sws.run_functions(cases_sws.Funcs(mi355.lib, "mi355_sws_"), luts, GOLD["seed"])
    bad = [k for k, want in GOLD["functions"].items() if sha(got[k]) != want]
    assert not bad, bad
    for seed in (3, 4):
        got = cases_sws.run_functions(cases_sws.Funcs(mi355.lib, "mi355_sws_"), luts, seed)
        want = cases_sws.run_functions(cases_sws.Funcs(oracle.lib, "oracle_sws_"), luts, seed)
        assert [k for k in want if want[k] != got[k]] == []


@pytest.mark.parametrize("name", list(S.CONFIGS))
def test_gpu_picture_matches_golden(mi355, name):
    ctx = S.load_context(name)
    got = S.product_backend(mi355).scale(ctx, S.picture(name), dst_pad=8)
    assert sha(got.tobytes()) == GOLD["pictures"][name]


@pytest.mark.parametrize("name", S.SMALL + ["hd_generic"])
def test_gpu_picture_matches_oracle_other_seed(mi355, oracle, name):
    ctx = S.load_context(name)
    planes = S.picture(name, seed=77, stride_pad=5)
    got = S.product_backend(mi355).scale(ctx, planes, dst_pad=3)
    assert (got == S.oracle_backend(oracle).scale(ctx, planes, dst_pad=3)).all()


def test_gpu_fate_pixfmt_rgb24_stage(mi355):
    ctx = S.load_context("cif_generic")
    rgb = S.product_backend(mi355).scale(ctx, S.fate_frame())
    assert sha(rgb.tobytes()) == GOLD["pictures"]["fate_pixfmt_rgb24_stage1"]


@pytest.mark.parametrize("name", ["down2_128x96", "special_70x50", "uhd_to_hd", "hd_special"])
def test_gpu_batched_frames(mi355, oracle, name):
    """Tier 2: several pictures per launch, device resident, unaligned strides"""
    ctx = S.load_context(name)
    pics = [S.picture(name, seed=s, stride_pad=2) for s in (1, 2, 3)]
    batch = S.DeviceBatch(mi355.lib, ctx, pics, 5, dst_pad=6)
    try:
        batch.run()
        for f in range(5):
            want = S.oracle_backend(oracle).scale(ctx, pics[f % 3 if f < 3 else f - 3])
            assert (batch.fetch(f) == want).all(), f
    finally:
        batch.close()
