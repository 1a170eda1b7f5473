"""One small pass of the hot path on the GPU, checked against the oracle (used by
__graft_entry__.smoke())."""
import cases_h264


def run(gpu, oracle):
    for group in ("idct", "qpel", "loopfilter"):
        got = cases_h264.run_group(gpu, group)
        want = cases_h264.run_group(oracle, group)
        assert got and set(got) <= set(want)
        bad = [k for k in got if got[k] != want[k]]
        assert not bad, bad[:10]
    try:
        import frame_cases
    except ImportError:
        return
    frame_cases.smoke(gpu, oracle)
