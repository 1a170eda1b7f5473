"""ctypes plumbing for the swscale part of the path: the public descriptor of include/mi355_sws.h,
interchangeable back-ends (oracle / product / emulated product / the reference's own libswscale),
saved contexts (filter banks + LUTs captured from the reference) and synthetic pictures."""
import ctypes as C
import os
import subprocess

import numpy as np

from rng import SplitMix64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "sws_contexts.npz")
HAVE_REFERENCE = os.path.isdir("/root/reference/libswscale")

i16p, i32p, u8p = C.POINTER(C.c_int16), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class Luts(C.Structure):
    _fields_ = [("y_table", C.c_uint8 * 1024), ("rV", C.c_int16 * 256), ("gU", C.c_int16 * 256),
                ("gV", C.c_int16 * 256), ("bU", C.c_int16 * 256)]


class Filter(C.Structure):
    _fields_ = [("coef", i16p), ("pos", i32p), ("size", C.c_int), ("n", C.c_int)]


class Desc(C.Structure):
    _fields_ = [("srcW", C.c_int), ("srcH", C.c_int), ("dstW", C.c_int), ("dstH", C.c_int),
                ("chrSrcW", C.c_int), ("chrSrcH", C.c_int), ("chrDstW", C.c_int), ("unscaled_special", C.c_int),
                ("hLum", Filter), ("hChr", Filter), ("vLum", Filter), ("vChr", Filter), ("luts", Luts)]


class SwsFrame(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("src_stride", C.c_int * 3), ("dst", C.c_void_p), ("dst_stride", C.c_int)]


INTS = ("srcW", "srcH", "dstW", "dstH", "chrSrcW", "chrSrcH", "chrDstW", "unscaled_special")
BANKS = ("hLum", "hChr", "vLum", "vChr")
LUTS = ("y_table", "rV", "gU", "gV", "bU")

# name: (srcW, srcH, dstW, dstH, bicubic, accurate_rnd, bitexact)
CONFIGS = {
    "special_64x48": (64, 48, 64, 48, 1, 0, 0),             # unscaled -> yuv2rgb_c_24_rgb
    "special_70x50": (70, 50, 70, 50, 1, 0, 0),             # width with 4 and 2 sample tails
    "generic_64x48": (64, 48, 64, 48, 1, 1, 1),             # accurate_rnd: generic path, chroma x2 vertically
    "generic_bilinear_72x40": (72, 40, 72, 40, 0, 1, 1),    # the same with two vertical chroma taps: yuv2rgb24_1_c (uvalpha below / above 2048 by row)
    "down2_128x96": (128, 96, 64, 48, 1, 1, 1),             # 2:1 bicubic, 8 taps
    "down_100x76": (100, 76, 64, 48, 1, 1, 1),              # odd ratio
    "down4_256x192": (256, 192, 64, 48, 1, 1, 1),           # 4:1: more than 8 taps per filter (the generic loops)
    "up2_bilinear": (64, 48, 128, 96, 0, 1, 1),             # 2-tap vertical filters: the _2 template
    "up_bicubic": (64, 48, 96, 80, 1, 1, 1),
    "cif_generic": (352, 288, 352, 288, 1, 1, 1),           # the shape of the FATE pixfmt tests
    # SURVEY.md §8d config 5 (full size)
    "hd_special": (1920, 1080, 1920, 1080, 1, 0, 0),
    "hd_generic": (1920, 1080, 1920, 1080, 1, 1, 1),
    "uhd_to_hd": (3840, 2160, 1920, 1080, 1, 1, 1),
}
SMALL = [k for k in CONFIGS if not k.startswith(("hd_", "uhd_"))]
SEED = 0x5A5


class Context:
    """A descriptor whose filter banks are owned by numpy arrays (no reference needed to use it)."""

    def __init__(self, ints, banks, luts):
        self.ints, self.banks, self.luts = dict(ints), banks, luts
        self.desc = d = Desc()
        for k in INTS:
            setattr(d, k, int(self.ints[k]))
        for k in BANKS:
            coef, pos = banks[k]
            coef, pos = np.ascontiguousarray(coef, np.int16), np.ascontiguousarray(pos, np.int32)
            banks[k] = (coef, pos)
            f = getattr(d, k)
            f.coef, f.pos = coef.ctypes.data_as(i16p), pos.ctypes.data_as(i32p)
            f.n, f.size = len(pos), (coef.size // len(pos) if len(pos) else 0)   # empty: the unscaled converter builds no filters
        for k in LUTS:
            arr = np.ascontiguousarray(luts[k])
            C.memmove(C.addressof(getattr(d.luts, k)), arr.ctypes.data, arr.nbytes)

    @staticmethod
    def from_desc(d):
        banks = {}
        for k in BANKS:
            f = getattr(d, k)
            if not f.coef or not f.pos:
                banks[k] = (np.zeros(0, np.int16), np.zeros(0, np.int32))
                continue
            banks[k] = (np.ctypeslib.as_array(f.coef, (f.n * f.size,)).copy(), np.ctypeslib.as_array(f.pos, (f.n,)).copy())
        luts = {k: np.frombuffer(bytes(getattr(d.luts, k)), np.uint8 if k == "y_table" else np.int16).copy() for k in LUTS}
        return Context({k: getattr(d, k) for k in INTS}, banks, luts)

    def arrays(self, prefix):
        out = {prefix + "/ints": np.array([self.ints[k] for k in INTS], np.int32)}
        for k in BANKS:
            out[prefix + "/" + k + "_coef"], out[prefix + "/" + k + "_pos"] = self.banks[k]
        for k in LUTS:
            out[prefix + "/" + k] = self.luts[k]
        return out

    @staticmethod
    def from_arrays(z, prefix):
        ints = dict(zip(INTS, (int(v) for v in z[prefix + "/ints"])))
        banks = {k: (z[prefix + "/" + k + "_coef"], z[prefix + "/" + k + "_pos"]) for k in BANKS}
        return Context(ints, banks, {k: z[prefix + "/" + k] for k in LUTS})


def load_context(name):
    with np.load(GOLDEN) as z:
        return Context.from_arrays(z, name)


def picture(name, seed=SEED, stride_pad=0):
    """uniform random yuv420p planes"""
    sw, sh = CONFIGS[name][:2]
    r = SplitMix64(seed * 7919 + sum(map(ord, name)))
    cw, ch = -(-sw // 2), -(-sh // 2)
    planes = [r.u8((sh, sw + stride_pad)), r.u8((ch, cw + stride_pad)), r.u8((ch, cw + stride_pad))]
    return planes


# ---- back-ends ---------------------------------------------------------------------------------
class Backend:
    """Uniform view over `<prefix>_sws_*` entry points taking the public descriptor."""

    def __init__(self, lib, prefix, name):
        self.lib, self.prefix, self.name = lib, prefix, name

    def fn(self, base):
        f = getattr(self.lib, self.prefix + base)
        f.restype = C.c_int if base in ("scale", "yuv2rgb_c_24_rgb") else None
        return f

    def scale(self, ctx, planes, dst_pad=0):
        d = ctx.desc
        out = np.full((d.dstH, d.dstW * 3 + dst_pad), 0x5A, np.uint8)
        src = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        strides = (C.c_int * 3)(*[p.strides[0] for p in planes])
        n = self._scale(ctx, src, strides, out)
        assert n == d.dstH, n
        return out

    def _scale(self, ctx, src, strides, out):
        return self.fn("scale")(C.byref(ctx.desc), src, strides, C.c_void_p(out.ctypes.data), C.c_int(out.strides[0]))


class ProductBackend(Backend):
    """mi355_sws_create(desc) + mi355_sws_scale(ctx, ...)"""

    def _scale(self, ctx, src, strides, out):
        self.lib.mi355_sws_create.restype = C.c_void_p
        h = self.lib.mi355_sws_create(C.byref(ctx.desc))
        assert h
        try:
            return self.fn("scale")(C.c_void_p(h), src, strides, C.c_void_p(out.ctypes.data), C.c_int(out.strides[0]))
        finally:
            self.lib.mi355_sws_destroy(C.c_void_p(h))


def oracle_backend(provider):
    return Backend(provider.lib, "oracle_sws_", "oracle")


def product_backend(provider):
    return ProductBackend(provider.lib, "mi355_sws_", provider.name)


# ---- the reference's own libswscale (only where /root/reference exists) ------------------------------
class Reference:
    def __init__(self):
        subprocess.run(["make", "-s", "_ref/libswsref.so"], cwd=os.path.join(ROOT, "oracle"), check=True)
        self.lib = lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libswsref.so"))
        lib.sws_getContext.restype = C.c_void_p
        lib.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
        lib.sws_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.sws_freeContext.argtypes = [C.c_void_p]
        lib.ref_sws_describe.argtypes = [C.c_void_p, C.c_void_p]
        self.name = "ref"

    def open(self, name, dst_fmt=1):
        sw, sh, dw, dh, bic, acc, bitexact = CONFIGS[name]
        flags = self.lib.ref_sws_flags_word(bic, acc, bitexact)
        c = self.lib.sws_getContext(sw, sh, self.lib.ref_pix_fmt(0), dw, dh, self.lib.ref_pix_fmt(dst_fmt), flags, None, None, None)
        assert c
        return c

    def close(self, c):
        self.lib.sws_freeContext(c)

    def context(self, name):
        c = self.open(name)
        d = Desc()
        assert self.lib.ref_sws_describe(c, C.byref(d)) == 0
        ctx = Context.from_desc(d)
        self.close(c)
        return ctx

    def scale(self, name, planes, dst_pad=0):
        sw, sh, dw, dh = CONFIGS[name][:4]
        c = self.open(name)
        out = np.full((dh, dw * 3 + dst_pad), 0x5A, np.uint8)
        src = (C.c_void_p * 4)(*[p.ctypes.data for p in planes], None)
        strides = (C.c_int * 4)(*[p.strides[0] for p in planes], 0)
        dst = (C.c_void_p * 4)(out.ctypes.data, None, None, None)
        dstrides = (C.c_int * 4)(out.strides[0], 0, 0, 0)
        n = self.lib.sws_scale(c, src, strides, 0, sh, dst, dstrides)
        assert n == dh, n
        self.close(c)
        return out


def reference():
    return Reference() if HAVE_REFERENCE else None


# ---- the FATE pin (tests/fate/pixfmt.mak, fate-run.sh:236-246 `pixfmt_conversion`) ----------------
FATE_FRAME = os.path.join(ROOT, "tests", "golden", "sws_vsynth1_00.npz")


def fate_frame():
    """frame 0 of the reference's synthetic test clip (tests/videogen.c), yuv420p 352x288"""
    with np.load(FATE_FRAME) as z:
        return [z["y"], z["u"], z["v"]]


def make_fate_frame():
    """run the reference's own tests/videogen.c (compiled in place into oracle/_ref) and parse 00.pgm"""
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "videogen")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-O1", "-w", "-I", "/root/reference", "-o", exe, "/root/reference/tests/videogen.c"], check=True)
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([exe, d + "/"], check=True, stdout=subprocess.DEVNULL)
        raw = open(os.path.join(d, "00.pgm"), "rb").read()
    magic, w, h, mx = raw.split(None, 4)[:4]
    assert magic == b"P5" and int(mx) == 255
    w, h = int(w), int(h)
    body = np.frombuffer(raw[len(raw) - w * h:], np.uint8).reshape(h, w)     # pgmyuv: luma, then U|V side by side
    H = h * 2 // 3
    y = body[:H].copy()
    c = body[H:]
    return [y, c[:, :w // 2].copy(), c[:, w // 2:].copy()]


def fate_chain_md5(ref, rgb):
    """second half of pixfmt_conversion: rgb24 -> yuv444p by the reference, md5 of the raw frame"""
    import hashlib
    lib = ref.lib
    h, w = rgb.shape[0], rgb.shape[1] // 3
    flags = lib.ref_sws_flags_word(1, 1, 1)
    c = lib.sws_getContext(w, h, lib.ref_pix_fmt(1), w, h, lib.ref_pix_fmt(2), flags, None, None, None)
    out = [np.zeros((h, w), np.uint8) for _ in range(3)]
    src = (C.c_void_p * 4)(rgb.ctypes.data, None, None, None)
    ss = (C.c_int * 4)(rgb.strides[0], 0, 0, 0)
    dst = (C.c_void_p * 4)(*[o.ctypes.data for o in out], None)
    ds = (C.c_int * 4)(w, w, w, 0)
    assert lib.sws_scale(c, src, ss, 0, h, dst, ds) == h
    lib.sws_freeContext(c)
    return hashlib.md5(b"".join(o.tobytes() for o in out)).hexdigest()


# ---- Tier 2: a batch of pictures resident in device memory -----------------------------------------
class DeviceBatch:
    """nframes pictures (the first `len(pictures)` distinct, the rest device-side copies) + output
    surfaces + the device array of mi355_sws_frame, through the C ABI's memory helpers."""

    def __init__(self, lib, ctx, pictures, nframes, src_pad=0, dst_pad=0):
        self.lib, self.ctx, self.n = lib, ctx, nframes
        lib.mi355_malloc.restype = C.c_void_p
        lib.mi355_malloc.argtypes = [C.c_size_t]
        lib.mi355_free.argtypes = [C.c_void_p]
        for f in ("mi355_memcpy_h2d", "mi355_memcpy_d2h", "mi355_memcpy_d2d"):
            getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mi355_sws_create.restype = C.c_void_p
        d = ctx.desc
        self.bufs = []
        G = len(pictures)
        self.strides = [pictures[0][p].shape[1] for p in range(3)]
        self.psz = [pictures[0][p].size for p in range(3)]
        self.src = []
        for p in range(3):
            base = self.alloc(nframes * self.psz[p] + 64)
            host = np.ascontiguousarray(np.stack([pic[p] for pic in pictures]))
            lib.mi355_memcpy_h2d(base, host.ctypes.data, host.nbytes)
            done = G
            while done < nframes:
                k = min(done, nframes - done)
                lib.mi355_memcpy_d2d(base + done * self.psz[p], base, k * self.psz[p])
                done += k
            self.src.append(base)
        self.dst_stride = (d.dstW * 3 + dst_pad + 3) & ~3
        self.dsz = self.dst_stride * d.dstH
        self.dst = self.alloc(nframes * self.dsz + 64)
        arr = (SwsFrame * nframes)()
        for f in range(nframes):
            for p in range(3):
                arr[f].src[p] = self.src[p] + f * self.psz[p]
                arr[f].src_stride[p] = self.strides[p]
            arr[f].dst = self.dst + f * self.dsz
            arr[f].dst_stride = self.dst_stride
        self.d_frames = self.alloc(C.sizeof(arr))
        lib.mi355_memcpy_h2d(self.d_frames, C.addressof(arr), C.sizeof(arr))
        self.handle = lib.mi355_sws_create(C.byref(d))
        assert self.handle

    def alloc(self, n):
        p = self.lib.mi355_malloc(n)
        assert p
        self.bufs.append(p)
        return p

    def run(self, stream=None):
        self.lib.mi355_sws_scale_frames_dev(C.c_void_p(self.handle), C.c_void_p(self.d_frames), self.n, C.c_void_p(stream))

    def fetch(self, f):
        d = self.ctx.desc
        out = np.empty((d.dstH, self.dst_stride), np.uint8)
        self.lib.mi355_sync(None)
        self.lib.mi355_memcpy_d2h(out.ctypes.data, self.dst + f * self.dsz, out.nbytes)
        return out[:, :d.dstW * 3]

    def close(self):
        self.lib.mi355_sws_destroy(C.c_void_p(self.handle))
        for p in self.bufs:
            self.lib.mi355_free(p)
        self.bufs = []
