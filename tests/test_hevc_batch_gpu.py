"""GPU: the batched HEVC entry points (include/mi355_hevc_batch.h) vs the oracle, bit-exact, 8/9/10 bit."""
import pytest

import hevc_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", (8, 9, 10))
@pytest.mark.parametrize("kind", list(hevc_batch.CHECKS))
def test_gpu_hevc_batches_match_oracle(mi355, oracle, kind, bd):
    for seed in (0x265 + bd, 77):
        assert hevc_batch.CHECKS[kind](mi355, oracle, bd, seed=seed) > 0


def test_gpu_hevc_residual_many_jobs(mi355, oracle):
    """more jobs than the machine holds at once"""
    assert hevc_batch.check_residual(mi355, oracle, 10, seed=5, cells=(48, 64)) == 48 * 64


def test_gpu_config3_chain_full_size_picture(mi355, oracle):
    """BASELINE config 3 at its full size: one 3840x2160 10-bit picture through every batched stage, compared with
    the oracle's table functions called in the same order (about 0.4 M jobs)"""
    import hevc_config3
    assert hevc_config3.check(mi355, oracle, 3840, 2160, 10, seed=0x265) > 0


def test_gpu_levels_launch_wait_that_runs_out_is_reported(mi355):
    hevc_batch.levels_wait_expiry_in_subprocess("mi355")
