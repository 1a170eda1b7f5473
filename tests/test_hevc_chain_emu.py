"""CPU (emulated kernels), where /root/reference exists: the chain bench.py measures for BASELINE config 3 (tools/hevc_chain.py: edge
emulation, fused MC + prediction, 32x32 transforms, picture-level deblocking, SAO of every CTB) against the REFERENCE's own functions
run over the same parameters (oracle/ref_hevc_chain.c in oracle/_ref/libhevcfilterref.so), every sample of the deblocked and of the
SAO output pictures — at sizes with whole and with ragged CTBs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/libavcodec"), reason="needs the reference's sources (/root/reference)")
@pytest.mark.parametrize("filter_fused", (True, False))
@pytest.mark.parametrize("width,height,bd", ((256, 192, 10), (320, 208, 10), (192, 144, 8)))
def test_measured_hevc_chain_matches_the_reference_functions_emulated(emu, width, height, bd, filter_fused):
    """filter_fused: deblocking + SAO of a coding tree block in one workgroup (mi355_hevc_filter_ctbs_dev) / the picture-level deblocking launches and the SAO launch"""
    import hevc_chain
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libhevcfilterref.so"], check=True)
    assert hevc_chain.check_against_reference(emu.lib, pictures=2, width=width, height=height, bd=bd, filter_fused=filter_fused) > 0     # some windows crossed a border
