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
