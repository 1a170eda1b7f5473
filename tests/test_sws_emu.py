"""CPU: the swscale kernels compiled against the SIMT emulator (test tooling) reproduce the oracle and
the reference's golden vectors through the C ABI of include/mi355_sws.h."""
import hashlib
import json
import os

import pytest

import cases_sws
import sws_support as S

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sws_ref_sha1.json")))


def sha(b):
    return hashlib.sha1(b).hexdigest()[:20]


def test_emulated_inner_loops_match_golden(emu):
    luts = S.load_context("down2_128x96").desc.luts
    got = cases_sws.run_functions(cases_sws.Funcs(emu.lib, "mi355_sws_"), luts, GOLD["seed"])
    bad = [k for k, want in GOLD["functions"].items() if sha(got[k]) != want]
    assert not bad, bad


def test_emulated_c24_slices_match_oracle(emu, oracle):
    luts = S.load_context("special_64x48").desc.luts
    got = cases_sws.run_functions(cases_sws.Funcs(emu.lib, "mi355_sws_"), luts, 3)
    want = cases_sws.run_functions(cases_sws.Funcs(oracle.lib, "oracle_sws_"), luts, 3)
    assert [k for k in want if want[k] != got[k]] == []


@pytest.mark.parametrize("name", S.SMALL)
def test_emulated_picture_matches_golden_and_oracle(emu, oracle, name):
    ctx = S.load_context(name)
    planes = S.picture(name)
    got = S.product_backend(emu).scale(ctx, planes, dst_pad=8)
    assert sha(got.tobytes()) == GOLD["pictures"][name]
    planes = S.picture(name, seed=9, stride_pad=5)
    assert (S.product_backend(emu).scale(ctx, planes, dst_pad=3) == S.oracle_backend(oracle).scale(ctx, planes, dst_pad=3)).all()
