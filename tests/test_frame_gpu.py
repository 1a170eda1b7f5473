"""GPU: Tier-2 batched reconstruction + deblocking through the C ABI vs the oracle, bit-exact."""
import numpy as np
import pytest

import frame_cases
import h264_frames as HF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_gpu(mi355, oracle, name):
    frame_cases.run_case(mi355, oracle, name)


def test_frame_pipeline_gpu_is_deterministic(mi355):
    fs = HF.synth_frames(nframes=4, mb_w=20, mb_h=12, seed=99, mix="mixed", intra_frac=0.1, refs="smooth", coef_b=6)
    outs = []
    for _ in range(2):
        d = HF.DeviceFrames(mi355, fs)
        d.decode()
        outs.append(d.fetch(d.dst))
        d.free()
    for p in range(3):
        assert np.array_equal(outs[0][p], outs[1][p])
