"""CPU: the batched HEVC entry points (include/mi355_hevc_batch.h) on the emulated kernels vs the oracle."""
import pytest

import hevc_batch


@pytest.mark.parametrize("bd", (8, 10))
@pytest.mark.parametrize("kind", list(hevc_batch.CHECKS))
def test_emulated_hevc_batches_match_oracle(emu, oracle, kind, bd):
    assert hevc_batch.CHECKS[kind](emu, oracle, bd, seed=0x265 + bd) > 0
