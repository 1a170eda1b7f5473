"""GPU: BASELINE config 3 as bench.py measures it (tools/hevc_chain.py) at its real size — 3840x2160, 10 bit, two pictures —
against the reference's own functions (oracle/_ref/libhevcfilterref.so, built HERE by __graft_entry__.build(); /root/reference
is not read on the GPU box): every sample of the deblocked pictures and of the SAO output."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("filter_fused", (True, False))
@pytest.mark.parametrize("width,height,bd", ((3840, 2160, 10), (320, 208, 10), (256, 192, 8), (3840, 2160, 8)))
def test_measured_hevc_chain_matches_the_reference_functions_gpu(mi355, width, height, bd, filter_fused):
    """filter_fused: deblocking + SAO of a coding tree block in one workgroup (mi355_hevc_filter_ctbs_dev: what bench.py's config-3 point runs) / the picture-level
    deblocking launches and the SAO launch"""
    import hevc_chain
    if hevc_chain.ref_library() is None:
        pytest.fail("oracle/_ref/libhevcfilterref.so missing: __graft_entry__.build() makes it where /root/reference exists")
    assert hevc_chain.check_against_reference(mi355.lib, pictures=2, width=width, height=height, bd=bd, filter_fused=filter_fused) > 0
