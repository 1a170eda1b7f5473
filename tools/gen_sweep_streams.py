"""Developer tool (needs /root/reference for the CAVLC tables): writes N random H.264 streams of the formats the second kernel set takes (9 / 10 bit,
4:2:2, 4:4:4 at 10 bit, transform bypass, MBAFF) into build/streams/sweep_*.samples — build/ is git-ignored but travels to the GPU box, where
tools/run_sweep_streams.sh decodes each with _ref/h264_bridge_gpu twice (bridge, plain) and compares.  usage: python tools/gen_sweep_streams.py [seed [count]]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_h264_streams as M
T = M.load_tables()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
out = os.path.join(ROOT, 'build', 'streams')
os.makedirs(out, exist_ok=True)
for f in os.listdir(out):
    if f.startswith('sweep_'): os.remove(os.path.join(out, f))
n = 0
for it in range(N):
    mbaff = rng.random() < 0.4
    fmt = rng.choice(((1, 8), (2, 8), (3, 8), (1, 10), (1, 10), (2, 10), (3, 10), (1, 9)))
    if mbaff:
        kw = dict(mb_w=rng.randrange(3, 14), mb_h=2 * rng.randrange(1, 6), chroma_idc=fmt[0], depth=fmt[1], seed=rng.randrange(1 << 30), nslices=rng.randrange(1, 5),
                  deblock_idc=rng.choice((-1, 0, 0, 1, 2)), weighted=bool(rng.randrange(2)), nrefs=rng.randrange(1, 4), npics=rng.randrange(3, 9), far=rng.choice((9, 20, 40)),
                  t8x8=bool(rng.randrange(2)), sparse=rng.choice((1.0, 0.5)))
        cls = M.MbaffStream
    else:
        if fmt == (1, 8): fmt = (2, 8)
        kw = dict(mb_w=rng.randrange(3, 14), mb_h=rng.randrange(2, 10), chroma_idc=fmt[0], depth=fmt[1], seed=rng.randrange(1 << 30), nslices=rng.randrange(1, 6),
                  deblock_idc=rng.choice((0, 0, 1, 2)), weighted=bool(rng.randrange(2)), nrefs=rng.randrange(1, 5), npics=rng.randrange(4, 11), far=rng.choice((9, 20, 40)),
                  bmode=rng.randrange(4), t8x8=bool(rng.randrange(2)), cip=bool(rng.randrange(2)), mixed=bool(rng.randrange(2)), paff=rng.random() < 0.25,
                  scaling=rng.random() < 0.3, sparse=rng.choice((1.0, 0.5)), skip=rng.choice((0.15, 0.5)), lossless=rng.random() < 0.2)
        if kw['lossless']: kw['weighted'] = False
        if kw['paff']:
            kw['mb_h'] += kw['mb_h'] & 1; kw['bmode'] = 0
            if kw['deblock_idc'] == 2: kw['deblock_idc'] = 0      # field pictures with idc 2: the reference is not consistent with itself (inconsistency (b))
        cls = M.Stream
    try:
        units = cls(T, 'sweep', **kw).build()
    except Exception as e:
        continue
    M.write_samples(os.path.join(out, 'sweep_%03d_%s%d_%d%s.samples' % (it, 'mbaff_' if mbaff else '', fmt[0], fmt[1], '_lossless' if kw.get('lossless') else '')), units)
    n += 1
print(n, 'streams in', out)
