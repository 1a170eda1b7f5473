"""GPU: the reference's own sws_scale() bound to the real library (oracle/_ref/libswsref_gpu.so = the reference's libswscale
+ contrib/libav/mi355_sws_glue.c linked with --wrap=ff_getSwsFunc,--wrap=ff_yuv2rgb_get_func_ptr against
libav_amd/libmi355dsp.so, built HERE by __graft_entry__.build(); /root/reference is not read on the GPU box).  Whole
pictures take the fused kernel, the inner-loop form (MI355_SWS_LINES=1) forwards line by line; both must give the golden
pictures of the plain reference (tests/golden/sws_ref_sha1.json)."""
import hashlib
import json
import os

import pytest

import sws_support as S
from test_sws_tier1_reference import bind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "sws_ref_sha1.json")))
LIB = os.path.join(ROOT, "oracle", "_ref", "libswsref_gpu.so")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bound(mi355):
    if not os.path.exists(LIB):
        pytest.fail("oracle/_ref/libswsref_gpu.so missing: __graft_entry__.build() makes it where /root/reference exists")
    ref = bind(S.Reference.__new__(S.Reference), LIB)
    ref.name = "ref+mi355"
    return ref


@pytest.mark.parametrize("name", sorted(k for k in GOLD["pictures"] if k in S.CONFIGS))
def test_reference_sws_scale_whole_pictures_on_the_gpu(bound, name, monkeypatch):
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    before = bound.lib.ref_sws_pictures()
    out = bound.scale(name, S.picture(name), dst_pad=8)
    assert bound.lib.ref_sws_pictures() == before + 1
    assert (out[:, -8:] == 0x5A).all()
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]


@pytest.mark.parametrize("name", [k for k in S.SMALL if not k.startswith("special")][:4])
def test_reference_sws_scale_inner_loops_on_the_gpu(bound, name, monkeypatch):
    monkeypatch.setenv("MI355_SWS_LINES", "1")
    before = bound.lib.ref_sws_tier1_calls()
    out = bound.scale(name, S.picture(name), dst_pad=8)
    assert bound.lib.ref_sws_tier1_calls() > before
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]
