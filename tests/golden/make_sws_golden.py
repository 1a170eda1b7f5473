"""Regenerate the swscale pins from the reference's OWN libswscale objects (oracle/_ref/libswsref.so,
built in place by oracle/Makefile; needs /root/reference):
  sws_contexts.npz   filter banks (initFilter) + yuv->rgb LUTs the reference built for every
                     configuration in sws_support.CONFIGS — inputs of the hot path, needed where
                     the reference is absent (GPU box);
  sws_ref_sha1.json  sha1 of the reference's output for every inner loop case (cases_sws.py) and for
                     one seeded picture per configuration.
Run:  python3 tests/golden/make_sws_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases_sws  # noqa: E402
import sws_support as S  # noqa: E402


def sha(b):
    return hashlib.sha1(b).hexdigest()[:20]


def main():
    ref = S.reference()
    assert ref, "needs /root/reference"
    arrays, pictures = {}, {}
    for name in S.CONFIGS:
        ctx = ref.context(name)
        arrays.update(ctx.arrays(name))
        pictures[name] = sha(ref.scale(name, S.picture(name), dst_pad=8).tobytes())
    np.savez_compressed(S.GOLDEN, **arrays)
    # FATE pin: frame 0 of tests/videogen.c -> rgb24 (this path) -> yuv444p (reference) must give the md5
    # stored in the reference's tests/ref/pixfmt/rgb24
    fy, fu, fv = S.make_fate_frame()
    np.savez_compressed(S.FATE_FRAME, y=fy, u=fu, v=fv)
    rgb = ref.scale("cif_generic", [fy, fu, fv])
    fate_md5 = open("/root/reference/tests/ref/pixfmt/rgb24").read().split()[0]
    assert S.fate_chain_md5(ref, rgb) == fate_md5
    pictures["fate_pixfmt_rgb24_stage1"] = sha(rgb.tobytes())
    rf = cases_sws.RefFuncs(ref)
    funcs = {k: sha(v) for k, v in cases_sws.run_functions(rf, rf.luts).items()}
    json.dump({"seed": S.SEED, "functions": funcs, "pictures": pictures, "luts": sha(bytes(rf.luts)),
               "fate_pixfmt_rgb24_md5": fate_md5},
              open(os.path.join(HERE, "sws_ref_sha1.json"), "w"), indent=1, sort_keys=True)
    print("sws: %d contexts, %d function cases, %d pictures" % (len(S.CONFIGS), len(funcs), len(pictures)))


if __name__ == "__main__":
    main()
