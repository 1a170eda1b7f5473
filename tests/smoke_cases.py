"""One small pass of the hot path on the GPU, checked against the oracle (used by
__graft_entry__.smoke()): Tier-1 tables (three groups), the batched H.264 picture pipeline in its three forms (one launch
per band, the small-batch loop filter, pictures of different geometry in one call, sparse coefficient fetch), the HEVC
picture-level loop filter and intra_pred wrapper, and one 10-bit H.264 table group against the reference-made goldens."""
import hashlib
import json
import os

import numpy as np

import cases_h264


def run(gpu, oracle):
    for group in ("idct", "qpel", "loopfilter"):
        got = cases_h264.run_group(gpu, group)
        want = cases_h264.run_group(oracle, group)
        assert got and set(got) <= set(want)
        bad = [k for k in got if got[k] != want[k]]
        assert not bad, bad[:10]
    import frame_cases
    frame_cases.smoke(gpu, oracle)
    frame_cases.run_case(gpu, oracle, "wide_mixed")                 # 37 macroblocks wide: the multi-band loop filter form
    frame_cases.run_case(gpu, oracle, "b_mixed", sparse=True)
    assert frame_cases.run_mixed_batch(gpu, oracle) >= 4
    # HEVC: picture-level deblocking and the intra_pred wrapper against their oracles
    import hevc_filter_cases as HC
    import hevc_intra_cases as IC
    oracle.lib.oracle_hevc_deblock_picture.restype = None
    want, _ = HC.run_host(oracle.lib.oracle_hevc_deblock_picture, "p10_64")
    got, _ = HC.run_device(gpu.lib, "p10_64", npics=2)
    assert all(np.array_equal(w, g) for pic in got for w, g in zip(want, pic))
    oracle.lib.oracle_hevc_intra_pred_blocks.restype = None
    want, _ = IC.run_host(oracle.lib.oracle_hevc_intra_pred_blocks, "i8_cip")
    got, _ = IC.run_device(gpu.lib, "i8_cip", npics=2)
    assert all(np.array_equal(w, g) for pic in got for w, g in zip(want, pic))
    # 10-bit H.264 tables: no CPU restatement exists, the checker is the golden file made by the reference's own code
    import cases_h264_hbd as HB
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h264dsp_hbd_ref_sha1.json")))["10"]
    got = HB.run_group(gpu, "qpel", 10)
    assert got and all(hashlib.sha1(v).hexdigest()[:20] == gold["qpel:" + k] for k, v in got.items())
