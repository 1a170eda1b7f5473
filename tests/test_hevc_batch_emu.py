"""CPU: the batched HEVC entry points (include/mi355_hevc_batch.h) on the emulated kernels vs the oracle."""
import pytest

import hevc_batch


@pytest.mark.parametrize("bd", (8, 10))
@pytest.mark.parametrize("kind", list(hevc_batch.CHECKS))
def test_emulated_hevc_batches_match_oracle(emu, oracle, kind, bd):
    assert hevc_batch.CHECKS[kind](emu, oracle, bd, seed=0x265 + bd) > 0


@pytest.mark.parametrize("bd", (8, 10))
def test_emulated_config3_chain_small_picture(emu, oracle, bd):
    """MC -> pred -> residual -> edges (V, H) -> SAO of one small picture, stage outputs feeding the next stage"""
    import hevc_config3
    assert hevc_config3.check(emu, oracle, 256, 192, bd, seed=3) > 0


def test_emulated_levels_launch_wait_that_runs_out_is_reported():
    hevc_batch.levels_wait_expiry_in_subprocess("emu")
