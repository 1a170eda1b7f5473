"""Random swscale contexts (yuv420p -> rgb24: picture sizes 2..400 in both directions, odd sizes, bicubic / bilinear, accurate
rounding on / off) through the reference's sws_scale() twice: its own code (oracle/_ref/libswsref.so) and the same library with
contrib/libav/mi355_sws_glue.c bound in, product kernels on the SIMT emulator (oracle/_ref/libswsref_tier1.so).  Pictures compared
byte for byte; the line says whether the product took the picture or the glue left the context to the reference.
Not a test of the suite: a sweep to run after touching the glue or sws.hip.  usage: python tools/sws_sweep.py [seed [count]]"""
import ctypes as C
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import providers
import sws_support as S

providers.emu()                                                # builds tests/_emu (the emulated product library the glue links)
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libswsref.so", "_ref/libswsref_tier1.so"], check=True)
plain = S.Reference()
hooked = S.Reference.__new__(S.Reference)
hooked.lib = lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libswsref_tier1.so"))
lib.sws_getContext.restype = C.c_void_p
lib.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
lib.sws_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.sws_freeContext.argtypes = [C.c_void_p]
lib.ref_sws_pictures.restype = C.c_ulong
os.environ.pop("MI355_SWS_LINES", None)

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = taken = 0
for it in range(N):
    def dim():
        return rng.choice((rng.randrange(2, 40), rng.randrange(16, 200), rng.randrange(100, 400)))
    sw, sh = dim(), dim()
    kind = rng.choice(('same', 'free', 'free', 'half', 'double'))
    dw, dh = {'same': (sw, sh), 'free': (dim(), dim()), 'half': (max(2, sw // 2), max(2, sh // 2)), 'double': (2 * sw, 2 * sh)}[kind]
    cfg = (sw, sh, dw, dh, rng.randrange(2), rng.randrange(2), rng.randrange(2))
    name = "sweep_%d" % it
    S.CONFIGS[name] = cfg
    planes = S.picture(name, seed=rng.randrange(1 << 20), stride_pad=rng.choice((0, 0, 5, 16)))
    try:
        a = plain.scale(name, planes, dst_pad=8)
        before = lib.ref_sws_pictures()
        b = hooked.scale(name, planes, dst_pad=8)
    except AssertionError as e:
        print(it, 'SKIP (the reference refuses the context)', cfg, repr(e)[:80])
        continue
    took = lib.ref_sws_pictures() - before
    taken += took
    # the picture itself; what lies right of it is compared apart: the reference writes pixels in pairs (yuv2rgb_write, output.c:
    # `for (i = 0; i < ((dstW + 1) >> 1); i++)`), so an odd width gets one pixel more than the picture has, the product writes dstW
    same = bool((a[:, :3 * dw] == b[:, :3 * dw]).all())
    ref_spill, our_spill = bool((a[:, 3 * dw:] != 0x5A).any()), bool((b[:, 3 * dw:] != 0x5A).any())
    print(it, 'OK' if same and not our_spill else 'MISMATCH', cfg, 'product' if took else 'reference',
          '(the reference writes past an odd width)' if ref_spill else '')
    a, b = a[:, :3 * dw], b[:, :3 * dw]
    same = same and not our_spill
    if not same:
        bad += 1
        ys, xs = np.nonzero(a != b)
        print('    first differences at rows %d..%d, bytes %d..%d (%d bytes)' % (ys.min(), ys.max(), xs.min(), xs.max(), len(ys)))
print('bad', bad, 'taken by the product', taken)
