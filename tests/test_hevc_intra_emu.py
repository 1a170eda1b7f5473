"""CPU: the product's HEVC intra_pred wrapper (mi355_hevc_intra_pred_blocks_dev) under the SIMT emulator vs the oracle
(oracle_hevc_intra.c, pinned to the reference's hevcpred_template.c) and the reference-made golden sha1s."""
import json

import numpy as np
import pytest

import hevc_intra_cases as IC
from test_oracle_hevc_filter import digest
from test_oracle_hevc_intra import GOLD


@pytest.mark.parametrize("name", list(IC.CASES))
def test_intra_pred_blocks_emulated(emu, oracle, name):
    oracle.lib.oracle_hevc_intra_pred_blocks.restype = None
    want, _ = IC.run_host(oracle.lib.oracle_hevc_intra_pred_blocks, name)
    outs, _ = IC.run_device(emu.lib, name, npics=2)
    for got in outs:
        for c in range(3):
            assert np.array_equal(want[c], got[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((want[c] != got[c]).sum()))
        assert digest(got) == json.load(open(GOLD))[name]


@pytest.mark.parametrize("name", list(IC.CASES))
def test_intra_blocks_with_their_residual_in_one_launch_emulated(emu, name):
    """mi355_hevc_intra_recon_blocks_dev (prediction + the block's transform unit, one launch per dependency level) = the two
    launches it replaces: mi355_hevc_intra_pred_blocks_dev, then mi355_hevc_residual_batch_dev over the same units (each of the
    two is pinned to the reference on its own: tests above, test_hevc_batch_emu.py)"""
    split, _ = IC.run_device(emu.lib, name, npics=2, residual="split")
    fused, _ = IC.run_device(emu.lib, name, npics=2, residual="fused")
    plain, _ = IC.run_device(emu.lib, name, npics=1)
    for a, b in zip(split, fused):
        for c in range(3):
            assert np.array_equal(a[c], b[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((a[c] != b[c]).sum()))
    assert any(not np.array_equal(split[0][c], plain[0][c]) for c in range(3))          # the units did change the pictures
