"""CPU: the product's picture-level HEVC deblocking driver (mi355_hevc_deblock_pictures_dev) under the SIMT emulator vs the
oracle (oracle_hevc_filter.c, pinned to the reference's hevc_filter.c) and the reference-made golden sha1s."""
import json
import os

import numpy as np
import pytest

import hevc_filter_cases as HC
from test_oracle_hevc_filter import GOLD, digest


@pytest.mark.parametrize("name", list(HC.CASES))
def test_deblock_pictures_emulated(emu, oracle, name):
    oracle.lib.oracle_hevc_deblock_picture.restype = None
    want, _ = HC.run_host(oracle.lib.oracle_hevc_deblock_picture, name)
    outs, _ = HC.run_device(emu.lib, name, npics=2)
    for got in outs:
        for c in range(3):
            assert np.array_equal(want[c], got[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((want[c] != got[c]).sum()))
        assert digest(got) == json.load(open(GOLD))["cases"][name]


import hashlib  # noqa: E402

import hevc_bs_cases as BC  # noqa: E402


@pytest.mark.parametrize("name", list(BC.CASES))
def test_boundary_strengths_emulated(emu, oracle, name):
    ov, oh, _ = BC.run_oracle(oracle.lib, name)
    res, c = BC.run_device(emu.lib, name, npics=2)
    mv, mh = BC.grid_mask(c)
    for v, h in res:
        assert np.array_equal(v[mv], ov[mv]) and np.array_equal(h[mh], oh[mh])
        assert (v[~mv] == 0xEE).all() and (h[~mh] == 0xEE).all()          # nothing outside the grid entries is written
        v2, h2 = np.where(mv, v, 0).astype(np.uint8), np.where(mh, h, 0).astype(np.uint8)
        assert hashlib.sha1(v2.tobytes() + h2.tobytes()).hexdigest()[:20] == json.load(open(GOLD))["bs_cases"][name]
