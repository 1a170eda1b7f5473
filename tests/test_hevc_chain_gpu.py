"""GPU: BASELINE config 3 as bench.py measures it (tools/hevc_chain.py) at its real size — 3840x2160, 10 bit, two pictures —
against the reference's own functions (oracle/_ref/libhevcfilterref.so, built HERE by __graft_entry__.build(); /root/reference
is not read on the GPU box): every sample of the deblocked pictures and of the SAO output."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width,height", ((3840, 2160), (320, 208)))
def test_measured_hevc_chain_matches_the_reference_functions_gpu(mi355, width, height):
    import hevc_chain
    if hevc_chain.ref_library() is None:
        pytest.fail("oracle/_ref/libhevcfilterref.so missing: __graft_entry__.build() makes it where /root/reference exists")
    assert hevc_chain.check_against_reference(mi355.lib, pictures=2, width=width, height=height, bd=10) > 0
