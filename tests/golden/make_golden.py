#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the REFERENCE's own compiled objects.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
Each entry is sha1(output buffers) of one checkasm-style case of tests/cases_*.py
executed through oracle/_ref/libref.so, i.e. produced by the reference C code
itself.  The oracle (and through it the MI355X backend) must reproduce them.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import providers  # noqa: E402


def digest(results):
    return {k: hashlib.sha1(v).hexdigest()[:20] for k, v in results.items()}


def main():
    ref = providers.ref()
    assert ref is not None, "needs /root/reference"
    import cases_h264
    out = {"seed": 0x264, "cases": digest(cases_h264.run_all(ref, 0x264))}
    with open(os.path.join(HERE, "h264dsp_ref_sha1.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("h264dsp:", len(out["cases"]), "cases")
    try:
        import cases_hevc
        out = {"seed": 0x265, "cases": digest(cases_hevc.run_all(ref, 0x265))}
        with open(os.path.join(HERE, "hevcdsp_ref_sha1.json"), "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        print("hevcdsp:", len(out["cases"]), "cases")
    except ImportError:
        pass
    # picture-level HEVC deblocking driver: the reference's own hevc_filter.c compiled in place
    import ctypes as C2
    import subprocess
    import hevc_filter_cases as HC
    root = os.path.dirname(os.path.dirname(HERE))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "_ref/libhevcfilterref.so"], check=True)
    flib = C2.CDLL(os.path.join(root, "oracle", "_ref", "libhevcfilterref.so"))
    flib.ref_hevc_deblock_picture.restype = C2.c_int
    cases = {}
    for name in HC.CASES:
        planes, _ = HC.run_host(flib.ref_hevc_deblock_picture, name)
        h = hashlib.sha1()
        for pl in planes:
            h.update(pl.tobytes())
        cases[name] = h.hexdigest()[:20]
    import hevc_bs_cases as BC
    bs_cases = {}
    for name in BC.CASES:
        v, h, _ = BC.run_reference(flib, name)
        bs_cases[name] = hashlib.sha1(v.tobytes() + h.tobytes()).hexdigest()[:20]
    with open(os.path.join(HERE, "hevc_filter_ref_sha1.json"), "w") as f:
        json.dump({"cases": cases, "bs_cases": bs_cases}, f, indent=0, sort_keys=True)
    print("hevc_filter:", len(cases), "+", len(bs_cases), "cases")
    # HEVC intra_pred wrapper: the reference's own HEVCPredContext.intra_pred[] (hevcpred_template.c) compiled in place
    import hevc_intra_cases as IC
    icases = {}
    for name in IC.CASES:
        planes, _ = IC.run_host(flib.ref_hevc_intra_pred_blocks, name)
        h = hashlib.sha1()
        for pl in planes:
            h.update(pl.tobytes())
        icases[name] = h.hexdigest()[:20]
    with open(os.path.join(HERE, "hevc_intra_ref_sha1.json"), "w") as f:
        json.dump(icases, f, indent=0, sort_keys=True)
    print("hevc_intra:", len(icases), "cases")
    # 9 / 10-bit H.264 tables: the reference's BIT_DEPTH 9 / 10 template instantiations (no CPU restatement exists)
    import cases_h264_hbd as HB
    hbd = {str(bd): digest(HB.run_all(ref, bd)) for bd in (9, 10)}
    with open(os.path.join(HERE, "h264dsp_hbd_ref_sha1.json"), "w") as f:
        json.dump(hbd, f, indent=0, sort_keys=True)
    print("h264dsp_hbd:", {k: len(v) for k, v in hbd.items()}, "cases")
    # struct layout of the pointer tables as the reference headers define them
    import ctypes as C
    buf = C.create_string_buffer(8192)
    ref.lib.ref_layout.restype = C.c_int
    n = ref.lib.ref_layout(buf, 8192)
    layout = dict(line.split("=") for line in buf.raw[:n].decode().split("\n") if line)
    with open(os.path.join(HERE, "abi_layout_ref.json"), "w") as f:
        json.dump({k: int(v) for k, v in layout.items()}, f, indent=0, sort_keys=True)
    print("layout:", len(layout), "entries")


if __name__ == "__main__":
    main()
