"""GPU: mi355_hevc_intra_pred_blocks_dev vs the oracle and the golden sha1s made by the reference's intra_pred[]."""
import json

import numpy as np
import pytest

import hevc_intra_cases as IC
from test_oracle_hevc_filter import digest
from test_oracle_hevc_intra import GOLD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(IC.CASES))
def test_intra_pred_blocks_gpu(mi355, oracle, name):
    oracle.lib.oracle_hevc_intra_pred_blocks.restype = None
    want, _ = IC.run_host(oracle.lib.oracle_hevc_intra_pred_blocks, name)
    outs, _ = IC.run_device(mi355.lib, name, npics=3)
    for got in outs:
        for c in range(3):
            assert np.array_equal(want[c], got[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((want[c] != got[c]).sum()))
        assert digest(got) == json.load(open(GOLD))[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(IC.CASES))
def test_intra_blocks_with_their_residual_in_one_launch_gpu(mi355, name):
    """prediction + the block's transform unit in one launch (mi355_hevc_intra_recon_blocks_dev) = the two launches it replaces"""
    split, _ = IC.run_device(mi355.lib, name, npics=2, residual="split")
    fused, _ = IC.run_device(mi355.lib, name, npics=2, residual="fused")
    for a, b in zip(split, fused):
        for c in range(3):
            assert np.array_equal(a[c], b[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((a[c] != b[c]).sum()))
