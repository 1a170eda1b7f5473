"""Random HEVC streams (tests/golden/make_hevc_streams.py with random parameters: tiles with uniform / explicit grids, wavefronts,
dependent slice segments, out-of-order picture groups, 8 / 9 / 10 bit, CTB 16 / 32 / 64, PCM, bypass, weights ...) through the
reference's decoder twice — plain, and with the reconstruction + filter bridges on the SIMT emulator (oracle/_ref/hevc_bridge_emu) —
and the two outputs compared.  Not a test of the suite (the streams are not kept): a sweep to run after touching the bridges.
usage: python tools/hevc_stream_sweep.py [seed [count]]"""
import sys, os, random, subprocess, hashlib, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_hevc_streams as M

TMP = tempfile.mkdtemp(prefix='hevc_sweep_')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for it in range(N):
    log2_ctb = rng.choice((4, 4, 5, 6))
    w = rng.choice((72, 104, 136, 168, 200)); h = rng.choice((56, 72, 88, 104, 136))
    cw, ch = -(-w >> log2_ctb), -(-h >> log2_ctb)
    kw = dict(seed=rng.randrange(1 << 30), w=w, h=h, log2_ctb=log2_ctb, log2_max_tb=min(5, log2_ctb), bd=rng.choice((8, 8, 9, 10)),
              sao=rng.choice((0, 1, 2)), slices=rng.randrange(1, 5), across=rng.randrange(2), inter=rng.randrange(2),
              qp=rng.randrange(22, 40), qp_delta=rng.randrange(2), pcm=rng.randrange(2), bypass=rng.randrange(2), tskip=rng.randrange(2),
              weighted=rng.randrange(2), cip=rng.randrange(2), dbf_offsets=rng.choice(((0, 0), (2, -1), (-2, 3))), dep=rng.randrange(2))
    kw['pictures'] = rng.randrange(3, 8) if kw['inter'] else 2
    if kw['cip']:
        # constrained intra prediction next to a slice or tile edge: when the neighbour above-left is intra but lies across the edge,
        # the reference's intra_pred reads a corner sample it never set (hevcpred_template.c:157-160 start at index 0, :187-199 leave
        # top[-1] alone for an intra neighbour) — its output then depends on what the stack held.  Such streams have no reference result.
        kw['slices'] = 1
        kw['dep'] = 0
    if kw['pcm']: kw['pcm_lf_off'] = rng.randrange(2)
    if kw['inter']: kw['pyramid'] = rng.randrange(2)
    mode = rng.choice(('tiles', 'tiles', 'wpp', 'none')) if not kw['cip'] else rng.choice(('wpp', 'none'))
    if mode == 'tiles' and cw >= 2 and ch >= 2:
        nc, nr = rng.randrange(1, min(cw, 4) + 1), rng.randrange(1, min(ch, 3) + 1)
        if nc * nr > 1:
            kw['tiles'] = (nc, nr); kw['across_tiles'] = rng.randrange(2)
            if rng.randrange(2):
                def split(total, n):
                    cuts = sorted(rng.sample(range(1, total), n - 1))
                    return tuple(b - a for a, b in zip([0] + cuts, cuts + [total]))
                kw['tile_sizes'] = (split(cw, nc), split(ch, nr))
    elif mode == 'wpp' and cw >= 2:
        kw['wpp'] = 1
    try:
        g = M.Hevc('sweep', **kw)
        pk = g.build()
    except Exception as e:
        print('GEN FAIL', kw, repr(e)); bad += 1; continue
    path = TMP + '/s%d.samples' % it
    M.write_samples(path, pk)
    outs = []
    ok = True
    for plain in (True, False):
        env = dict(os.environ)
        for k in ("MI355_HEVC_RECON_PLAIN", "MI355_HEVC_LF_PLAIN"): env.pop(k, None)
        if plain: env["MI355_HEVC_RECON_PLAIN"] = env["MI355_HEVC_LF_PLAIN"] = "1"
        out = TMP + '/o%d_%d.yuv' % (it, plain)
        r = subprocess.run([ROOT + '/oracle/_ref/hevc_bridge_emu', path, out], capture_output=True, text=True, env=env, timeout=900)
        if r.returncode or r.stderr.strip():
            print('RUN FAIL plain=%s' % plain, kw, r.stderr[-400:]); ok = False; break
        j = json.loads(r.stdout.strip().splitlines()[-1])
        outs.append((hashlib.md5(open(out, 'rb').read()).hexdigest(), j))
    if not ok: bad += 1; continue
    same = outs[0][0] == outs[1][0]
    j = outs[1][1]
    on_dev = j['pictures_reconstructed_on_device'] == j['pictures_output'] == j['pictures_filtered_on_device'] == kw['pictures']
    print(it, 'OK' if same and on_dev else 'MISMATCH', {k: v for k, v in kw.items() if k in ('tiles', 'tile_sizes', 'across_tiles', 'wpp', 'dep', 'pyramid', 'bd', 'log2_ctb', 'slices', 'across', 'inter', 'sao')}, j['pictures_output'], j['reference_uploads'])
    if not (same and on_dev): bad += 1; print('   ', kw)
print('bad', bad)
