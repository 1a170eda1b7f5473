#!/usr/bin/env python3
"""Developer soak: random Tier-2 cases (geometry, partition mix, B / weighted / intra / 8x8 transform shares drawn per case)
through the GPU pipeline against the oracle, in the plain, sparse-fetch and forced multi-band forms.
usage (GPU box): python tools/soak_frames.py [cases] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import h264_frames as HF  # noqa: E402
import providers  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    gpu, orc = providers.mi355(), providers.oracle()
    bad = 0
    for k in range(n):
        r = np.random.default_rng(seed0 + k)
        kw = dict(nframes=int(r.integers(1, 4)), mb_w=int(r.integers(1, 60)), mb_h=int(r.integers(1, 30)), seed=seed0 + k,
                  mix=("p16", "mixed")[int(r.integers(0, 2))], intra_frac=float(r.choice([0.0, 0.05, 0.3, 1.0])),
                  bframes=bool(r.integers(0, 2)), weighted=int(r.integers(0, 3)), dct8_frac=float(r.choice([0.0, 0.3])),
                  mv_range=int(r.choice([16, 64, 300])), offsets=bool(r.integers(0, 2)), pcm_frac=float(r.choice([0.0, 0.05])),
                  refs=("noise", "smooth")[int(r.integers(0, 2))], coef_b=int(r.choice([4, 24, 200])))
        fs = HF.synth_frames(**kw)
        sparse = bool(r.integers(0, 2))
        if sparse:
            inter = (fs.mb["mb_type"] & 7) == 0
            pick = inter & (r.random(inter.shape) < 0.5)
            fs.mb["cbp"][pick] = 0
            fs.mb["nnz_mask"][pick] = 0
            fs.coef[pick] = 0
        recon_o, dst_o = HF.run_oracle(orc, fs)
        d = HF.DeviceFrames(gpu, fs, pad=int(r.choice([0, 0, 8, 24])))
        try:
            if sparse:
                d.decode_sparse()
            else:
                d.decode(per_level=bool(r.integers(0, 2)))
            recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
        finally:
            d.free()
        ok = all(np.array_equal(a, b) for a, b in zip(recon_o, recon_g)) and all(np.array_equal(a, b) for a, b in zip(dst_o, dst_g))
        if not ok:
            bad += 1
            print("MISMATCH", k, kw, "sparse" if sparse else "dense", flush=True)
    print("soak: %d cases, %d mismatches (MI355_DEBLOCK_FORM=%s)" % (n, bad, os.environ.get("MI355_DEBLOCK_FORM", "auto")))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
