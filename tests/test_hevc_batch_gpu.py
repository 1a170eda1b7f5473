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
