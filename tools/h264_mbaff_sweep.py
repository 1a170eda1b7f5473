"""Random MBAFF streams (tests/golden/make_h264_streams.py MbaffStream: macroblock pairs coded as frame or field macroblocks at random, I / P / B pictures,
intra macroblocks in P and B pictures, explicit and implicit weights, slices with their own filter mode, 4:2:0 / 4:2:2 / 4:4:4 at 8 and 10 bit) through the reference's
decoder twice: plain (MI355_BRIDGE_PLAIN) and with the Tier-2 bridge on the SIMT emulator (oracle/_ref/h264_bridge_emu), outputs compared.
A sweep to run after touching the MBAFF path.  usage: python tools/h264_mbaff_sweep.py [seed [count]]"""
import sys, os, random, subprocess, hashlib, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_h264_streams as M

TMP = tempfile.mkdtemp(prefix='h264_mbaff_sweep_')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = M.load_tables()
EXE = os.path.join(ROOT, 'oracle', '_ref', 'h264_bridge_emu')
bad = 0
for it in range(N):
    fmt = rng.choice(((1, 8), (1, 8), (2, 8), (3, 8), (1, 10), (2, 10), (3, 10)))
    kw = dict(mb_w=rng.randrange(3, 11), mb_h=2 * rng.randrange(1, 5), chroma_idc=fmt[0], depth=fmt[1], seed=rng.randrange(1 << 30),
              nslices=rng.randrange(1, 5), deblock_idc=rng.choice((-1, 0, 0, 1, 2)), weighted=bool(rng.randrange(2)), nrefs=rng.randrange(1, 4),
              npics=rng.randrange(3, 9), far=rng.choice((9, 20, 40)), t8x8=bool(rng.randrange(2)), cip=bool(rng.randrange(2)),
              sparse=rng.choice((1.0, 0.5)), bmode=rng.choice((0, 0, 1, 1, 2, 3)))
    try:
        units = M.MbaffStream(T, 'sweep', **kw).build()
    except Exception as e:
        print(it, 'GEN SKIP', repr(e)[:120], kw)
        continue
    path = os.path.join(TMP, 's%d.samples' % it)
    M.write_samples(path, units)
    variant = rng.choice(('default', 'lazy', 'threads2', 'direct'))
    outs = []
    for plain in (True, False):
        env = dict(os.environ)
        for k in ("MI355_BRIDGE_LAZY", "MI355_BRIDGE_DIRECT", "MI355_BRIDGE_PLAIN", "MI355_BRIDGE_SESSION", "MI355_BRIDGE_LINEAR"):
            env.pop(k, None)
        threads = 1
        if plain:
            env["MI355_BRIDGE_PLAIN"] = "1"
        else:
            if variant == 'lazy': env["MI355_BRIDGE_LAZY"] = "1"
            if variant == 'direct': env["MI355_BRIDGE_DIRECT"] = "1"
            if variant == 'threads2': threads = 2
        out = os.path.join(TMP, 'o%d_%d.yuv' % (it, plain))
        r = subprocess.run([EXE, path, out, str(threads), "1"], capture_output=True, text=True, env=env, timeout=1800)
        st = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        outs.append((hashlib.md5(open(out, 'rb').read()).hexdigest() if os.path.exists(out) else None, st[-1] if st else {}, r.stderr.strip(), r.returncode))
    if outs[0][2] or outs[0][3]:
        print(it, 'SKIP (the reference decoder rejects the stream: %s)' % outs[0][2].splitlines()[-1][-60:] if outs[0][2] else 'rc', kw)
        continue
    j = outs[1][1]
    same = outs[0][0] == outs[1][0]
    on_dev = j.get('pictures_on_device', 0) >= j.get('pictures_output', -1)
    verdict = 'OK' if same and on_dev else ('MISMATCH' if not same else 'NOT ON DEVICE')
    print(it, verdict, variant, 'on device %s/%s' % (j.get('pictures_on_device'), j.get('pictures_output')), {k: v for k, v in kw.items() if k not in ('seed', 'far', 'sparse')})
    if verdict != 'OK':
        bad += 1
        print('    ', kw, outs[1][2][-300:])
print('bad', bad)
