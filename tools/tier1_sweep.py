"""Random streams from the two writers of tests/golden/ through the reference's decoders with the Tier-1 pointer tables hooked
(oracle/_ref/h264_tier1_emu, hevc_tier1_emu: every table entry the hooks replace runs the product's per-call kernels on the SIMT
emulator; the HEVC run also with the picture-level filter bridge, hevc_lf_emu) against the same decoder with its tables untouched
(MI355_TIER1_PLAIN).  H.264 draws cover what Tier 2 does not take: 4:2:2, 9 / 10 bit, lossless, MBAFF.
Not a test of the suite.  usage: python tools/tier1_sweep.py [seed [count]]"""
import hashlib
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_h264_streams as H4
import make_hevc_streams as H5

TMP = tempfile.mkdtemp(prefix='tier1_sweep_')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = H4.load_tables()
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h264_tier1_emu", "_ref/hevc_tier1_emu", "_ref/hevc_lf_emu"], check=True)


def run(exe, path, out, plain, extra=()):
    env = dict(os.environ)
    for k in ("MI355_TIER1_PLAIN", "MI355_HEVC_LF_PLAIN", "MI355_HEVC_INTRA_DEVICE"):
        env.pop(k, None)
    if plain:
        env["MI355_TIER1_PLAIN"] = "1"
        env["MI355_HEVC_LF_PLAIN"] = "1"
    for k in extra:
        env[k] = "1"
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), path, out], capture_output=True, text=True, env=env, timeout=1800)
    lines = [l for l in r.stderr.splitlines() if l.strip()]
    ok = r.returncode == 0 and len(lines) == 1
    return hashlib.md5(open(out, 'rb').read()).hexdigest() if os.path.exists(out) else None, ok, (lines[-1] if lines else '')[-110:]


bad = 0
for it in range(N):
    if rng.random() < 0.6:
        fmt = rng.choice(((1, 8), (2, 8), (2, 10), (1, 10), (1, 9), (3, 8), (3, 10)))
        kind = rng.choice(('plain', 'plain', 'mbaff', 'lossless', 'paff'))
        kw = dict(mb_w=rng.randrange(3, 10), mb_h=rng.randrange(2, 8), chroma_idc=fmt[0], depth=fmt[1], seed=rng.randrange(1 << 30),
                  nslices=rng.randrange(1, 5), deblock_idc=rng.choice((-1, 0, 0, 1, 2)), weighted=bool(rng.randrange(2)), nrefs=rng.randrange(1, 4),
                  npics=rng.randrange(4, 9), bmode=rng.randrange(4), t8x8=bool(rng.randrange(2)), cip=bool(rng.randrange(2)),
                  scaling=rng.random() < 0.3, npps=rng.choice((1, 2)))
        cls = H4.Stream
        if kind == 'mbaff':
            cls = H4.MbaffStream
            kw.update(bmode=0, cip=False, npps=1, scaling=False)
            kw['mb_h'] += kw['mb_h'] & 1
        elif kind == 'lossless':
            kw.update(lossless=True, weighted=False, bmode=0)
        elif kind == 'paff':
            kw.update(paff=True, bmode=0)
            kw['mb_h'] += kw['mb_h'] & 1
            if kw['deblock_idc'] == 2:
                kw['deblock_idc'] = 0
        try:
            units = cls(T, 'sweep', **kw).build()
        except Exception as e:
            print(it, 'h264 GEN SKIP', kind, repr(e)[:80])
            continue
        path = os.path.join(TMP, 's%d.samples' % it)
        H4.write_samples(path, units)
        a = run('h264_tier1_emu', path, os.path.join(TMP, 'a.yuv'), True)
        if not a[1]:
            print(it, 'h264 SKIP (the reference rejects the stream)', kind, a[2]); continue
        b = run('h264_tier1_emu', path, os.path.join(TMP, 'b.yuv'), False)
        ok = b[1] and a[0] == b[0]
        print(it, 'h264', kind, fmt, 'OK' if ok else 'MISMATCH', b[2][-70:])
        if not ok:
            bad += 1; print('    ', kw)
    else:
        log2_ctb = rng.choice((4, 5, 6))
        kw = dict(seed=rng.randrange(1 << 30), w=rng.choice((72, 104, 136, 168)), h=rng.choice((56, 72, 104)), log2_ctb=log2_ctb, log2_max_tb=min(5, log2_ctb),
                  bd=rng.choice((8, 9, 10)), sao=rng.choice((0, 1, 2)), slices=rng.randrange(1, 4), across=rng.randrange(2), inter=rng.randrange(2),
                  qp=rng.randrange(22, 40), qp_delta=rng.randrange(2), pcm=rng.randrange(2), bypass=rng.randrange(2), tskip=rng.randrange(2),
                  weighted=rng.randrange(2), scaling=rng.randrange(2), dep=rng.randrange(2))
        kw['pictures'] = rng.randrange(3, 7) if kw['inter'] else 2
        if kw['inter']:
            kw['pyramid'] = rng.randrange(2)
        path = os.path.join(TMP, 's%d.samples' % it)
        H5.write_samples(path, H5.Hevc('sweep', **kw).build())
        a = run('hevc_tier1_emu', path, os.path.join(TMP, 'a.yuv'), True)
        b = run('hevc_tier1_emu', path, os.path.join(TMP, 'b.yuv'), False)
        c = run('hevc_lf_emu', path, os.path.join(TMP, 'c.yuv'), False)
        d = run('hevc_tier1_emu', path, os.path.join(TMP, 'd.yuv'), False, extra=("MI355_HEVC_INTRA_DEVICE",))
        ok = a[1] and b[1] and c[1] and d[1] and a[0] == b[0] == c[0] == d[0]
        print(it, 'hevc', {k: kw[k] for k in ('bd', 'log2_ctb', 'inter', 'sao', 'slices')}, 'OK' if ok else 'MISMATCH', b[2][-60:])
        if not ok:
            bad += 1; print('    ', kw, a, b, c, d)
print('bad', bad)
