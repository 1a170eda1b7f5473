"""Random H.264 streams (tests/golden/make_h264_streams.py with random parameters: picture sizes, slices, P / B with implicit / explicit /
no weights, 8x8 transforms, constrained intra prediction, field pairs, re-ordered lists, several parameter sets, scaling lists, gaps,
reference marking, 4:2:0 / 4:2:2 / 4:4:4 at 8, 9 and 10 bit, transform bypass — everything the Tier-2 bridge takes since round 4: the 8-bit 4:2:0 /
4:4:4 kernels, and the second kernel set for the rest) through the
reference's decoder twice: plain (MI355_BRIDGE_PLAIN) and with contrib/libav/mi355_h264_bridge.c on the SIMT emulator
(oracle/_ref/h264_bridge_emu; lazily finished pictures, the session facade and two decoder threads at random), outputs compared.
Not a test of the suite: a sweep to run after touching the bridge.  usage: python tools/h264_stream_sweep.py [seed [count]]"""
import sys, os, random, subprocess, hashlib, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_h264_streams as M

TMP = tempfile.mkdtemp(prefix='h264_sweep_')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = M.load_tables()
EXE = os.path.join(ROOT, 'oracle', '_ref', 'h264_bridge_emu')
bad = 0
for it in range(N):
    fmt = rng.choice(((1, 8), (1, 8), (3, 8), (2, 8), (2, 8), (1, 10), (1, 10), (2, 10), (1, 9), (3, 10)))
    # pictures at least three macroblocks wide: with a 16-byte chroma line the reference's two-reference weighted prediction keeps its Cb
    # and Cr intermediates in overlapping rows (h264_mb.c:407-409: tmp_cr = tmp_cb + 16, rows mb_uvlinesize apart)
    kw = dict(mb_w=rng.randrange(3, 12), mb_h=rng.randrange(2, 9), chroma_idc=fmt[0], depth=fmt[1], seed=rng.randrange(1 << 30),
              nslices=rng.randrange(1, 6), deblock_idc=rng.choice((-1, 0, 0, 1, 2)), weighted=bool(rng.randrange(2)), nrefs=rng.randrange(1, 5),
              npics=rng.randrange(4, 11), far=rng.choice((9, 20, 40)), bmode=rng.randrange(4), t8x8=bool(rng.randrange(2)),
              cip=bool(rng.randrange(2)), mixed=bool(rng.randrange(2)), paff=rng.random() < 0.3, reorder=rng.random() < 0.3,
              npps=rng.choice((1, 1, 3)), scaling=rng.random() < 0.3, gaps=rng.random() < 0.2, mmco=rng.random() < 0.2,
              sparse=rng.choice((1.0, 0.5)), skip=rng.choice((0.15, 0.5)), lossless=rng.random() < 0.15)
    if kw['lossless']:
        kw['weighted'] = False
    if kw['paff']:
        kw['mb_h'] += kw['mb_h'] & 1                  # field pairs: an even number of macroblock rows
        kw['bmode'] = 0                               # the writer's field pictures are I / P
        kw['reorder'] = kw['gaps'] = kw['mmco'] = False   # ... with default lists and marking
        # field pictures with disable_deblocking_filter_idc 2: the reference decides whether an intra macroblock's unfiltered
        # above-left border is swapped in from slice_table[mb_xy - 1 - mb_stride] (h264_mb.c:525-527) — the row of the OTHER field,
        # whose entries are whatever an earlier picture left: its output differs from its own idc 0 output on one-slice pictures
        # (-1 lets every slice draw its own idc, 2 among them: sweep 41 of round 4 found the same inconsistency that way — draws 4 and 37,
        # 4:4:4 field pictures; with idc 2 written as 0 on one-slice pictures the reference's own output becomes the bridge's)
        if kw['deblock_idc'] in (2, -1):
            kw['deblock_idc'] = 0
    if kw['mmco']:
        kw['nrefs'] = max(kw['nrefs'], 3)
    try:
        units = M.Stream(T, 'sweep', **kw).build()
    except Exception as e:
        print(it, 'GEN SKIP', repr(e)[:120], kw)
        continue
    path = os.path.join(TMP, 's%d.samples' % it)
    M.write_samples(path, units)
    variant = rng.choice(('default', 'lazy', 'session', 'threads2', 'lazy_direct'))
    outs = []
    ok = True
    for plain in (True, False):
        env = dict(os.environ)
        for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_PLAIN", "MI355_BRIDGE_SESSION", "MI355_BRIDGE_LINEAR"):
            env.pop(k, None)
        threads = 1
        if plain:
            env["MI355_BRIDGE_PLAIN"] = "1"
        else:
            if variant.startswith('lazy'): env["MI355_BRIDGE_LAZY"] = "1"
            if variant == 'lazy_direct': env["MI355_BRIDGE_DIRECT"] = "1"
            if variant == 'session' and fmt in ((1, 8),) and not kw['lossless']: env["MI355_BRIDGE_SESSION"] = "1"
            if variant == 'threads2': threads = 2
        out = os.path.join(TMP, 'o%d_%d.yuv' % (it, plain))
        r = subprocess.run([EXE, path, out, str(threads), "1"], capture_output=True, text=True, env=env, timeout=1800)
        if r.returncode:
            print(it, 'RUN FAIL plain=%s' % plain, kw, r.stderr[-400:]); ok = False; break
        st = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        # several decoder threads write one output file each (suffix .N): take them all
        datas = b""
        for n in range(threads):
            p = out if n == 0 else out + ".%d" % n
            if os.path.exists(p): datas += open(p, 'rb').read()
        outs.append((hashlib.md5(datas if plain or threads == 1 else open(out, 'rb').read()).hexdigest(), st[-1] if st else {}, r.stderr.strip()))
    if not ok: bad += 1; continue
    if outs[0][2]:
        # the plain decoder complains: a combination the writer does not produce valid streams for (field pairs with list
        # re-ordering / gaps / marking operations) — nothing to compare against
        print(it, 'SKIP (the reference decoder rejects the stream: %s)' % outs[0][2].splitlines()[-1][-60:])
        continue
    same = outs[0][0] == outs[1][0]
    j = outs[1][1]
    in_scope = True
    # field pictures count one by one on the device, pairs come out as one frame
    on_dev = j.get('pictures_on_device', 0) >= j.get('pictures_output', -1)
    verdict = 'OK' if same and (on_dev or not in_scope) else ('MISMATCH' if not same else 'NOT ON DEVICE')
    print(it, verdict, variant, 'on device %s/%s' % (j.get('pictures_on_device'), j.get('pictures_output')),
          {k: v for k, v in kw.items() if k not in ('seed', 'far', 'sparse', 'skip')})
    if verdict != 'OK':
        bad += 1
        print('    ', kw, outs[1][2][-300:])
print('bad', bad)
