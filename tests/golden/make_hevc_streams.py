#!/usr/bin/env python3
"""make_hevc_streams.py — TEST INFRASTRUCTURE: a small HEVC bitstream WRITER (CABAC) for streams no offline sample covers.

The reference tree holds no HEVC sample, so nothing ran this project's HEVC Tier-1 tables (hevcdsp / hevcpred hooks)
INSIDE the reference decoder.  This script writes syntactically valid HEVC streams from seeded random choices — the
coding quadtree, intra modes, transform trees, residuals (all scans, all transform sizes 4..32), SAO, deblocking control,
cu_qp_delta, transform skip, transquant bypass, several slices, 8 and 10 bit; with `inter`: P and B pictures (merge,
AMVP with random vector differences, all partition shapes, weighted prediction).  It is open loop: the writer never
reconstructs a picture; it only tracks what the SYNTAX depends on (coding-tree depth, skip flags, intra modes for the
most-probable-mode derivation and the scan choice).  What the streams decode to is defined by the reference decoder:
a stream is accepted when the reference decodes it without a single complaint, and the golden value is the md5 of that
plain decode (tests/golden/hevc_streams.json); tests then run the same decoder with the tables replaced.

The arithmetic coder's tables (range / state transition) and the context initialisation values are READ from the
reference's sources at generation time (libavcodec/cabac.c ff_h264_cabac_tables, hevc_cabac.c init_values /
num_bins_in_se) — nothing of them is stored here.  Syntax order and context selection follow the decoder:
hevcdec.c hls_slice_header :466, hls_sao_param :831, hls_residual_coding :902, hls_transform_tree :1364,
hls_coding_unit :2053, hls_coding_quadtree :2202; hevc_cabac.c :418-873; hevc_ps.c (parameter sets).

usage: python tests/golden/make_hevc_streams.py [--check]     (needs /root/reference and oracle/_ref/hevc_tier1_emu)
"""
import hashlib
import json
import os
import random
import re
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MI355_REFERENCE", "/root/reference")
OUT = HERE                                                    # small (1.5 - 4.5 KB each): committed, travel to the GPU box


# ---------------------------------------------------------------- tables read from the reference
def _ints(text):
    return [int(v, 0) for v in re.findall(r"-?\b(?:0x[0-9a-fA-F]+|\d+)\b", text)]


def _strip(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def load_tables():
    cab = _strip(open(os.path.join(REF, "libavcodec/cabac.c")).read())
    m = re.search(r"ff_h264_cabac_tables\[[^\]]*\]\s*=\s*\{(.*?)\};", cab, re.S)
    T = [v & 0xFF for v in _ints(m.group(1))]               # a uint8_t table written with negative literals
    assert len(T) >= 1280
    hc = _strip(open(os.path.join(REF, "libavcodec/hevc_cabac.c")).read())
    nb = _ints(re.search(r"num_bins_in_se\[\]\s*=\s*\{(.*?)\};", hc, re.S).group(1))
    body = re.search(r"init_values\[3\]\[HEVC_CONTEXTS\]\s*=\s*\{(.*?)\n\};", hc, re.S).group(1)
    body = body.replace("CNU", "154")
    vals = _ints(body)
    n = sum(nb)
    assert len(vals) == 3 * n, (len(vals), n)
    off, o = [], 0
    for b in nb:
        off.append(o)
        o += b
    return T, [vals[i * n:(i + 1) * n] for i in range(3)], off


(SAO_MERGE, SAO_TYPE, SAO_EO, SAO_BAND, SAO_ABS, SAO_SIGN, END_SLICE, SPLIT_CU, TQ_BYPASS, SKIP, QP_DELTA, PRED_MODE, PART_MODE,
 PCM, PREV_INTRA, MPM_IDX, REM_INTRA, CHROMA_MODE, MERGE_FLAG, MERGE_IDX, INTER_IDC, REF_L0, REF_L1, MVD_G0, MVD_G1, MVD_M2,
 MVD_SIGN, MVP_FLAG, NO_RESID, SPLIT_TT, CBF_LUMA, CBF_C, TSKIP, LAST_X, LAST_Y, LAST_XS, LAST_YS, SIG_CG, SIG_COEFF, GT1, GT2,
 REMAIN, SIGN) = range(43)


# ---------------------------------------------------------------- bits
class Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        for i in range(n - 1, -1, -1):
            self.b.append((v >> i) & 1)

    def ue(self, v):
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def trailing(self):
        self.b.append(1)
        while len(self.b) % 8:
            self.b.append(0)

    def align_one(self):                     # byte_alignment(): a one, then zeros
        self.trailing()

    def bytes(self):
        assert len(self.b) % 8 == 0
        out = bytearray()
        for i in range(0, len(self.b), 8):
            v = 0
            for k in self.b[i:i + 8]:
                v = (v << 1) | k
            out.append(v)
        return bytes(out)


def nal(nut, payload, tid=0):
    out = bytearray([nut << 1, tid + 1])
    z = 0
    for byte in payload:
        if z >= 2 and byte <= 3:
            out.append(3)
            z = 0
        out.append(byte)
        z = z + 1 if byte == 0 else 0
    return b"\x00\x00\x00\x01" + bytes(out)


# ---------------------------------------------------------------- the arithmetic encoder (H.265 9.3.4 / the inverse of cabac_functions.h)
class Cabac:
    def __init__(self, tables, bits):
        self.T, self.init, self.off = tables
        self.out = bits
        self.low, self.range, self.first, self.outstanding = 0, 510, 1, 0
        self.state = []

    def init_states(self, init_type, qp):
        self.state = []
        for v in self.init[init_type]:
            m = (v >> 4) * 5 - 45
            n = ((v & 15) << 3) - 16
            pre = 2 * (((m * min(max(qp, 0), 51)) >> 4) + n) - 127
            pre ^= pre >> 31
            if pre > 124:
                pre = 124 + (pre & 1)
            self.state.append(pre)

    def _put(self, b):
        if self.first:
            self.first = 0
        else:
            self.out.b.append(b)
        while self.outstanding:
            self.out.b.append(1 - b)
            self.outstanding -= 1

    def _renorm(self):
        while self.range < 256:
            if self.low < 256:
                self._put(0)
            elif self.low >= 512:
                self.low -= 512
                self._put(1)
            else:
                self.low -= 256
                self.outstanding += 1
            self.range <<= 1
            self.low <<= 1

    def enc(self, elem, inc, b):
        i = self.off[elem] + inc
        s = self.state[i]
        lps = self.T[512 + 2 * (self.range & 0xC0) + s]
        self.range -= lps
        if b == (s & 1):
            self.state[i] = self.T[1024 + 128 + s]
        else:
            self.low += self.range
            self.range = lps
            self.state[i] = self.T[1024 + 127 - s]
        self._renorm()

    def byp(self, b):
        self.low <<= 1
        if b:
            self.low += self.range
        if self.low >= 1024:
            self._put(1)
            self.low -= 1024
        elif self.low < 512:
            self._put(0)
        else:
            self.low -= 512
            self.outstanding += 1

    def restart(self):
        """after pcm_flag (a terminating bin equal to 1, flushed by term()) and the raw samples: 9.3.2.5 initialisation, contexts kept"""
        self.low, self.range, self.first, self.outstanding = 0, 510, 1, 0

    def byps(self, n, v):
        for i in range(n - 1, -1, -1):
            self.byp((v >> i) & 1)

    def term(self, b):
        self.range -= 2
        if b:
            self.low += self.range
            self.range = 2
            self._renorm()
            self._put((self.low >> 9) & 1)
            self.out.u(2, ((self.low >> 7) & 3) | 1)
        else:
            self._renorm()


# ---------------------------------------------------------------- scans (H.265 6.5.3-6.5.5)
def diag_scan(n):
    out, x, y, stop = [], 0, 0, False
    while not stop:
        while y >= 0:
            if x < n and y < n:
                out.append((x, y))
            y -= 1
            x += 1
        y, x = x, 0
        if len(out) >= n * n:
            stop = True
    return out


DIAG = {n: diag_scan(n) for n in (1, 2, 4, 8)}
HORIZ4 = [(i & 3, i >> 2) for i in range(16)]
HORIZ2 = [(0, 0), (1, 0), (0, 1), (1, 1)]
SCAN_DIAG, SCAN_HORIZ, SCAN_VERT = 0, 1, 2


def scan_tables(log2, scan_idx):
    """-> (sub-block order [(x_cg, y_cg)], position order inside a sub-block [(x, y)]) as hls_residual_coding picks them"""
    ncg = 1 << (log2 - 2)
    if scan_idx == SCAN_DIAG:
        return DIAG[ncg], DIAG[4]
    if scan_idx == SCAN_HORIZ:
        return (HORIZ2 if log2 == 3 else [(0, 0)]), HORIZ4
    return ([(y, x) for x, y in HORIZ2] if log2 == 3 else [(0, 0)]), [(y, x) for x, y in HORIZ4]


# ---------------------------------------------------------------- the stream
class Hevc:
    def __init__(self, name, seed, w=96, h=64, bd=8, log2_ctb=5, log2_min_cb=3, log2_min_tb=2, log2_max_tb=5, depth_intra=2,
                 depth_inter=2, sao=1, dbf_off=0, dbf_offsets=(0, 0), strong=1, qp=30, qp_delta=0, tskip=0, bypass=0, slices=1,
                 pictures=2, cb_off=0, cr_off=0, amp=1, inter=0, weighted=0, cip=0, density=0.35, scaling=0, across=1, sdh=0, pcm=0, pcm_lf_off=0, intra_frac=0.3, tiles=None, across_tiles=1, tile_sizes=None, pyramid=0, wpp=0, dep=0, vary_refs=0):
        self.__dict__.update(locals())
        self.rng = random.Random(seed)
        self.tables = load_tables()

    # ---- parameter sets (hevc_ps.c)
    def ptl(self, b):
        b.u(2, 0); b.u(1, 0); b.u(5, 1 if self.bd == 8 else 2)
        b.u(32, 0x60000000 if self.bd == 8 else 0x20000000)
        b.u(1, 1); b.u(1, 0); b.u(1, 0); b.u(1, 1)
        b.u(32, 0); b.u(11, 0); b.u(1, 0)
        b.u(8, 93)

    def vps(self):
        b = Bits()
        b.u(4, 0); b.u(2, 3); b.u(6, 0); b.u(3, 0); b.u(1, 1); b.u(16, 0xFFFF)
        self.ptl(b)
        b.u(1, 1); b.ue(4); b.ue(2); b.ue(0)
        b.u(6, 0); b.ue(0); b.u(1, 0); b.u(1, 0)
        b.trailing()
        return nal(32, b.bytes())

    def sps(self):
        b = Bits()
        b.u(4, 0); b.u(3, 0); b.u(1, 1)
        self.ptl(b)
        b.ue(0); b.ue(1); b.ue(self.w); b.ue(self.h); b.u(1, 0)
        b.ue(self.bd - 8); b.ue(self.bd - 8); b.ue(4)              # log2_max_poc_lsb 8
        b.u(1, 1); b.ue(4); b.ue(2); b.ue(0)
        b.ue(self.log2_min_cb - 3); b.ue(self.log2_ctb - self.log2_min_cb)
        b.ue(self.log2_min_tb - 2); b.ue(self.log2_max_tb - self.log2_min_tb)
        b.ue(self.depth_inter); b.ue(self.depth_intra)
        b.u(1, 1 if self.scaling else 0)
        if self.scaling:
            b.u(1, 0)                                                # default lists
        b.u(1, self.amp); b.u(1, 1 if self.sao else 0); b.u(1, 1 if self.pcm else 0)
        if self.pcm:                                                 # PCM samples one / two bits narrower than the pictures': put_pcm shifts
            b.u(4, self.bd - 2); b.u(4, self.bd - 3); b.ue(0); b.ue(min(self.log2_ctb, 5) - 3); b.u(1, self.pcm_lf_off)
        b.ue(0)                                                      # no short-term sets in the SPS: every slice carries its own
        b.u(1, 0); b.u(1, 0); b.u(1, self.strong); b.u(1, 0); b.u(1, 0)
        b.trailing()
        return nal(33, b.bytes())

    def pps(self):
        b = Bits()
        b.ue(0); b.ue(0); b.u(1, self.dep); b.u(1, 0); b.u(3, 0); b.u(1, self.sdh); b.u(1, 0)
        b.ue(1); b.ue(1)                                             # two references by default in both lists
        b.se(0); b.u(1, self.cip); b.u(1, self.tskip); b.u(1, 1 if self.qp_delta else 0)
        if self.qp_delta:
            b.ue(1)                                                  # diff_cu_qp_delta_depth
        b.se(self.cb_off); b.se(self.cr_off); b.u(1, 0)
        b.u(1, self.weighted); b.u(1, self.weighted); b.u(1, self.bypass); b.u(1, 1 if self.tiles else 0); b.u(1, self.wpp)
        if self.tiles:                                               # hevc_ps.c: columns, rows, uniform spacing or explicit sizes
            b.ue(self.tiles[0] - 1); b.ue(self.tiles[1] - 1)
            b.u(1, 0 if self.tile_sizes else 1)
            if self.tile_sizes:
                for v in self.tile_sizes[0][:-1]:
                    b.ue(v - 1)
                for v in self.tile_sizes[1][:-1]:
                    b.ue(v - 1)
            b.u(1, self.across_tiles)
        b.u(1, self.across)
        ctl = self.dbf_off or any(self.dbf_offsets)
        b.u(1, 1 if ctl else 0)
        if ctl:
            b.u(1, 0); b.u(1, self.dbf_off)
            if not self.dbf_off:
                b.se(self.dbf_offsets[0]); b.se(self.dbf_offsets[1])
        b.u(1, 0); b.u(1, 0); b.ue(0); b.u(1, 0); b.u(1, 0)
        b.trailing()
        return nal(34, b.bytes())

    # ---- one picture
    def picture(self, poc):
        """slice types: picture 0 an IDR; with `inter` the rest alternate P and B, each naming the two pictures before it"""
        lc = 1 << self.log2_ctb
        self.cw, self.ch = (self.w + lc - 1) >> self.log2_ctb, (self.h + lc - 1) >> self.log2_ctb
        n4w, n4h = self.cw * lc // 4, self.ch * lc // 4
        self.n4w = n4w
        self.depth = [[0] * n4w for _ in range(n4h)]               # coding-tree depth, 4x4 granularity
        self.ipm = [[1] * n4w for _ in range(n4h)]                 # intra mode (INTRA_DC where not intra)
        self.skipf = [[0] * n4w for _ in range(n4h)]
        nctb = self.cw * self.ch
        self.neg, self.pos = [poc - 1 - i for i in range(min(poc, 2))], []       # the reference picture set: before / after in output order
        if self.pyramid and poc:
            # groups of four in decoding order: the anchor (4 after the last), then the middle, then the two between — pictures
            # leave the decoder in another order than they enter it, and references lie on both sides
            g, k = divmod(poc - 1, 4)
            base = 4 * g
            poc = base + (4, 2, 1, 3)[k]
            self.neg, self.pos = {4: ([base], []), 2: ([base], [base + 4]), 1: ([base], [base + 2, base + 4]),
                                  3: ([base + 2, base], [base + 4])}[poc - base]
        self.poc = poc
        self.pcm_blocks = getattr(self, "pcm_blocks", [])
        idr = poc == 0 or not self.inter
        self.stype = 2 if idr else (1 if poc % 2 else 0)           # HEVC_SLICE_B 0, P 1, I 2
        if self.pyramid and not idr:
            self.stype = 1 if poc % 8 == 4 else 0                    # every other anchor a P picture
        self.nrefs = 0 if idr else len(self.neg) + len(self.pos)
        self.tile_scan()
        starts = sorted(set([0] + [self.rng.randrange(1, nctb) for _ in range(self.slices - 1)])) if nctb > 1 else [0]
        if self.tiles:
            # H.265 6.3.1: a slice holds whole tiles, or a tile whole slices — slice starts (positions in tile scan) are a
            # subset of the tile starts in one picture, a superset in the next
            tstarts = [t for t in range(nctb) if t == 0 or self.tile_id[self.ts2rs[t]] != self.tile_id[self.ts2rs[t - 1]]]
            if poc % 2:
                starts = sorted(set(starts) | set(tstarts))
            else:
                starts = sorted(set([0] + self.rng.sample(tstarts[1:], min(self.slices - 1, len(tstarts) - 1))))
        if self.wpp:
            # wavefronts: slices begin with a row here (the reference's decoder takes a row's contexts from the row above whatever
            # slice that lay in, hevc_cabac.c:397-411; with slices of whole rows that is what 9.3.1 says too)
            starts = sorted(set([0] + [self.rng.randrange(1, self.ch) * self.cw for _ in range(self.slices - 1)])) if self.ch > 1 else [0]
        out = b""
        self.n_slices = getattr(self, "n_slices", 0) + len(starts)
        self.n_ctus = getattr(self, "n_ctus", 0) + nctb
        for si, first in enumerate(starts):
            end = starts[si + 1] if si + 1 < len(starts) else nctb
            dependent = int(bool(self.dep and si and self.rng.random() < 0.6))     # a slice segment that continues the slice before it
            out += self.slice(poc, idr, first, end, nctb, dependent)
        return out

    def tile_scan(self):
        """H.265 6.5.1 (hevc_ps.c setup_pps): coding tree blocks in tile scan, the tile every block lies in"""
        nctb = self.cw * self.ch
        if not self.tiles:
            self.ts2rs, self.tile_id = list(range(nctb)), [0] * nctb
            return
        nc, nr = self.tiles
        if self.tile_sizes:
            cols, rows = self.tile_sizes
        else:
            cols = [(i + 1) * self.cw // nc - i * self.cw // nc for i in range(nc)]
            rows = [(i + 1) * self.ch // nr - i * self.ch // nr for i in range(nr)]
        assert sum(cols) == self.cw and sum(rows) == self.ch and min(cols) > 0 and min(rows) > 0
        self.ts2rs, self.tile_id = [], [0] * nctb
        y0 = 0
        for j, th in enumerate(rows):
            x0 = 0
            for i, tw in enumerate(cols):
                for y in range(y0, y0 + th):
                    for x in range(x0, x0 + tw):
                        self.ts2rs.append(y * self.cw + x)
                        self.tile_id[y * self.cw + x] = j * nc + i
                x0 += tw
            y0 += th

    def slice(self, poc, idr, first, end, nctb, dependent=0):
        r = self.rng
        b = Bits()
        b.u(1, 1 if first == 0 else 0)
        if idr:
            b.u(1, 0)
        b.ue(0)
        if first:
            if self.dep:
                b.u(1, dependent)
            b.u((nctb - 1).bit_length(), self.ts2rs[first])             # slice_segment_address: raster scan
        if dependent:
            # hls_slice_header (hevcdec.c): everything else is the slice's; the contexts go on from the end of the segment before
            # (ff_hevc_cabac_init: cabac_init_state only for independent segments)
            return self.slice_data(b, idr, first, end, self.c.state)
        self.slice_rs = self.ts2rs[first]
        b.ue(self.stype)
        if not idr:
            b.u(8, poc & 255)
            b.u(1, 0)                                                # the slice's own short-term set (hevc_ps.c ff_hevc_decode_short_term_rps)
            b.ue(len(self.neg)); b.ue(len(self.pos))
            at = poc
            for v in self.neg:
                b.ue(at - v - 1); b.u(1, 1)                          # delta_poc_s0_minus1, used
                at = v
            at = poc
            for v in self.pos:
                b.ue(v - at - 1); b.u(1, 1)                          # delta_poc_s1_minus1, used
                at = v
        sao_l = sao_c = 0
        if self.sao:
            sao_l, sao_c = r.randrange(2) if self.sao == 1 else 1, r.randrange(2) if self.sao == 1 else 1
            b.u(1, sao_l); b.u(1, sao_c)
        if not idr:
            b.u(1, 1)                                                # num_ref_idx_active_override
            if self.vary_refs:
                # every slice its own number of active references: the slices of a picture then use DIFFERENT lists (what the filter
                # bridge's one-table-per-picture boundary-strength pass cannot rate: it must fall back to the decoder's strengths)
                self.nact = r.randrange(1, self.nrefs + 1)
            else:
                self.nact = self.nrefs
            b.ue(self.nact - 1)
            if self.stype == 0:
                b.ue(self.nact - 1)
                b.u(1, 0)                                            # mvd_l1_zero_flag
            if self.weighted:
                self.pred_weights(b)
            self.max_merge = r.randrange(1, 6)
            b.ue(5 - self.max_merge)
        sqp = self.qp + r.randrange(-3, 4)
        b.se(sqp - 26)
        if self.across and (sao_l or sao_c or not self.dbf_off):
            b.u(1, r.randrange(2))
        self.sqp, self.sao_l, self.sao_c = sqp, sao_l, sao_c
        return self.slice_data(b, idr, first, end, None)

    def slice_data(self, b, idr, first, end, states):
        sqp = self.sqp
        d = Bits()                                                   # slice_segment_data(): starts on a byte
        c = Cabac(self.tables, d)
        if states is None or (self.wpp and self.cw == 1) or \
           (self.tiles and self.tile_id[self.ts2rs[first]] != self.tile_id[self.ts2rs[first - 1]]):
            c.init_states(2 - self.stype, sqp)                       # also a dependent segment that opens a tile (hevc_cabac.c:381-385)
        elif self.wpp:
            c.state = list(self.wpp_saved)                           # a dependent segment at the start of a row: the row above decides
        else:
            c.state = list(states)
        self.c = c
        self.sao_tab = getattr(self, "sao_tab", {})
        first_rs = self.slice_rs                                     # of the SLICE: a dependent segment's neighbours in the segments before it count
        subs = []                                                    # byte positions where the tiles / rows after the first begin
        for ts in range(first, end):
            addr = self.ts2rs[ts]
            self.first_in_slice, self.addr = first_rs, addr
            rx, ry = addr % self.cw, addr // self.cw
            # hls_decode_neighbour (hevcdec.c:2255-2300): in the slice (distance in raster scan, as the decoder has it) and in the tile
            in_slice = addr - first_rs
            self.left_ok = rx > 0 and in_slice > 0 and self.tile_id[addr] == self.tile_id[addr - 1]
            self.up_ok = ry > 0 and in_slice >= self.cw and self.tile_id[addr] == self.tile_id[addr - self.cw]
            self.sao_syntax(rx, ry)
            self.quadtree(rx << self.log2_ctb, ry << self.log2_ctb, self.log2_ctb, 0)
            if self.wpp and rx == 1:
                self.wpp_saved = list(c.state)                       # 9.3.2.2 / ff_hevc_save_states: after the second block of a row
            c.term(1 if ts == end - 1 else 0)
            new_row = self.wpp and ts + 1 < end and self.ts2rs[ts + 1] % self.cw == 0
            if ts + 1 < end and (new_row or self.tile_id[self.ts2rs[ts + 1]] != self.tile_id[addr]):
                c.term(1)                                            # end_of_subset_one_bit, byte_alignment(): the flush's last bit is the one
                while len(d.b) % 8:
                    d.b.append(0)
                subs.append(len(d.b) // 8)
                c.restart()
                if new_row and self.cw > 1:
                    c.state = list(self.wpp_saved)
                else:
                    c.init_states(2 - self.stype, sqp)
        while len(d.b) % 8:
            d.b.append(0)
        data = d.bytes()
        if not (self.tiles or self.wpp):
            b.align_one()
            return nal(19 if idr else 1, b.bytes() + data)
        # entry points (7.4.7.1): sizes of the subsets in bytes of the NAL unit, emulation prevention bytes included — those depend
        # on the header in front, so: write, count, write again until the sizes stand
        sizes = [q - p for p, q in zip([0] + subs, subs + [len(data)])]
        for _ in range(8):
            hb = Bits()
            hb.b = list(b.b)
            hb.ue(len(subs))
            if subs:
                hb.ue(31)
                for v in sizes[:-1]:
                    hb.u(32, v - 1)
            hb.align_one()
            head = hb.bytes()
            unit = nal(19 if idr else 1, head + data)
            # position of every payload byte in the escaped unit
            pos, z, k = [], 0, 6
            for byte in head + data:
                if z >= 2 and byte <= 3:
                    k += 1
                    z = 0
                pos.append(k)
                k += 1
                z = z + 1 if byte == 0 else 0
            pos.append(k)
            bounds = [len(head) + q for q in [0] + subs + [len(data)]]
            now = [pos[bounds[i + 1]] - pos[bounds[i]] for i in range(len(bounds) - 1)]
            if now == sizes:
                return unit
            sizes = now
        raise RuntimeError("entry points do not settle")

    def pred_weights(self, b):
        r = self.rng
        ld, cd = r.randrange(0, 8), r.randrange(0, 8)
        b.ue(ld)                                                     # luma_log2_weight_denom
        b.se(cd - ld)                                                # delta_chroma_log2_weight_denom
        for _l in range(2 if self.stype == 0 else 1):
            lf = [r.randrange(2) for _ in range(self.nact)]
            cf = [r.randrange(2) for _ in range(self.nact)]
            for f in lf:
                b.u(1, f)
            for f in cf:
                b.u(1, f)
            for i in range(self.nact):
                if lf[i]:
                    b.se(r.randrange(-20, 21)); b.se(r.randrange(-30, 31))
                if cf[i]:
                    for _ in range(2):
                        b.se(r.randrange(-20, 21)); b.se(r.randrange(-30, 31))

    # ---- SAO (hls_sao_param)
    def sao_syntax(self, rx, ry):
        if not (self.sao_l or self.sao_c):
            return
        c, r = self.c, self.rng
        ml = mu = 0
        if rx > 0 and self.left_ok:
            ml = int(r.random() < 0.2)
            c.enc(SAO_MERGE, 0, ml)
        if ry > 0 and not ml and self.up_ok:
            mu = int(r.random() < 0.2)
            c.enc(SAO_MERGE, 0, mu)
        if ml or mu:
            return
        typ = 0
        for ci in range(3):
            if not (self.sao_c if ci else self.sao_l):
                continue
            if ci < 2:
                typ = r.choice((0, 1, 2, 2))                         # not applied, band, edge
                c.enc(SAO_TYPE, 0, 1 if typ else 0)
                if typ:
                    c.byp(typ - 1)
            if not typ:
                continue
            length = (1 << (min(self.bd, 10) - 5)) - 1
            offs = [min(length, int(r.expovariate(0.6))) for _ in range(4)]
            for o in offs:
                for _ in range(o):
                    c.byp(1)
                if o < length:
                    c.byp(0)
            if typ == 1:
                for o in offs:
                    if o:
                        c.byp(r.randrange(2))
                c.byps(5, r.randrange(32))
            elif ci < 2:
                c.byps(2, r.randrange(4))

    # ---- coding quadtree / coding unit
    def nb(self, tab, x, y):
        return tab[y >> 2][x >> 2]

    def fill(self, tab, x, y, size, v):
        for j in range(y >> 2, min((y + size) >> 2, len(tab))):
            row = tab[j]
            for i in range(x >> 2, min((x + size) >> 2, self.n4w)):
                row[i] = v

    def quadtree(self, x0, y0, log2, depth):
        c, r = self.c, self.rng
        size = 1 << log2
        lc = (1 << self.log2_ctb) - 1
        if x0 + size <= self.w and y0 + size <= self.h and log2 > self.log2_min_cb:
            inc = 0
            if (self.left_ok or (x0 & lc)) and self.nb(self.depth, x0 - 1, y0) > depth:
                inc += 1
            if (self.up_ok or (y0 & lc)) and self.nb(self.depth, x0, y0 - 1) > depth:
                inc += 1
            split = int(r.random() < (0.75 if log2 > 4 else 0.5))
            c.enc(SPLIT_CU, inc, split)
        else:
            split = int(log2 > self.log2_min_cb)
        if self.qp_delta and log2 >= self.log2_ctb - 1:
            self.qp_coded = 0
        if split:
            h = size >> 1
            for (x, y) in ((x0, y0), (x0 + h, y0), (x0, y0 + h), (x0 + h, y0 + h)):
                if x < self.w and y < self.h:
                    self.quadtree(x, y, log2 - 1, depth + 1)
        else:
            self.ct_depth = depth
            self.coding_unit(x0, y0, log2)
            self.fill(self.depth, x0, y0, size, depth)

    def coding_unit(self, x0, y0, log2):
        c, r = self.c, self.rng
        size = 1 << log2
        lc = (1 << self.log2_ctb) - 1
        self.bypass_cu = 0
        if self.bypass:
            self.bypass_cu = int(r.random() < 0.25)
            c.enc(TQ_BYPASS, 0, self.bypass_cu)
        intra, skip = 1, 0
        if self.stype != 2:
            inc = 0
            if (self.left_ok or (x0 & lc)) and self.nb(self.skipf, x0 - 1, y0):
                inc += 1
            if (self.up_ok or (y0 & lc)) and self.nb(self.skipf, x0, y0 - 1):
                inc += 1
            skip = int(r.random() < 0.2)
            c.enc(SKIP, inc, skip)
            self.fill(self.skipf, x0, y0, size, skip)
        if skip:
            self.merge_idx()
            self.fill(self.ipm, x0, y0, size, 1)
            return
        if self.stype != 2:
            intra = int(r.random() < self.intra_frac)       # share of intra coding units in P / B slices
            c.enc(PRED_MODE, 0, intra)
        part = 0                                                     # 2Nx2N
        if intra:
            if log2 == self.log2_min_cb:
                nxn = int(r.random() < 0.5) if log2 > self.log2_min_tb else 0
                c.enc(PART_MODE, 0, 1 - nxn)
                part = 3 if nxn else 0
            if part == 0 and self.pcm and 3 <= log2 <= min(self.log2_ctb, 5):
                pcm = int(r.random() < 0.25)
                c.term(pcm)
                if pcm:
                    b = c.out
                    while len(b.b) % 8:
                        b.b.append(0)                                # pcm_alignment_zero_bit
                    luma = [r.randrange(1 << (self.bd - 1)) for _ in range(size * size)]
                    for v in luma:
                        b.u(self.bd - 1, v)
                    for _ in range(size * size // 2):
                        b.u(self.bd - 2, r.randrange(1 << (self.bd - 2)))
                    c.restart()
                    self.fill(self.ipm, x0, y0, size, 1)
                    self.pcm_blocks.append((self.poc, x0, y0, size, luma))
                    return
            self.intra_pu(x0, y0, log2, part == 3)
            self.max_depth = self.depth_intra + (1 if part == 3 else 0)
            self.intra_cu, self.intra_split = 1, part == 3
            self.transform_tree(x0, y0, x0, y0, log2, 0, 0, 0, 0)
            return
        self.fill(self.ipm, x0, y0, size, 1)
        part = self.part_mode(log2)
        h, q = size >> 1, size >> 2
        pbs = {0: [(0, 0, size, size)], 1: [(0, 0, size, h), (0, h, size, h)], 2: [(0, 0, h, size), (h, 0, h, size)],
               3: [(0, 0, h, h), (h, 0, h, h), (0, h, h, h), (h, h, h, h)],
               4: [(0, 0, size, q), (0, q, size, size - q)], 5: [(0, 0, size, size - q), (0, size - q, size, q)],
               6: [(0, 0, q, size), (q, 0, size - q, size)], 7: [(0, 0, size - q, size), (size - q, 0, q, size)]}[part]
        merged = 0
        for (_px, _py, pw, ph) in pbs:
            merged = self.prediction_unit(pw, ph)
        root = 1
        if not (part == 0 and merged):
            root = int(r.random() < 0.7)
            c.enc(NO_RESID, 0, root)
        if root:
            self.max_depth = self.depth_inter
            self.intra_cu, self.intra_split, self.part = 0, False, part
            self.cur_mode = self.mode_c = 1
            self.transform_tree(x0, y0, x0, y0, log2, 0, 0, 0, 0)

    def part_mode(self, log2):
        """ff_hevc_part_mode_decode, inter: 0 2Nx2N, 1 2NxN, 2 Nx2N, 3 NxN, 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N"""
        c, r = self.c, self.rng
        if log2 == self.log2_min_cb:
            choices = [0, 1, 2] + ([3] if log2 > 3 else [])
            p = r.choice(choices)
            if p == 0:
                c.enc(PART_MODE, 0, 1)
            else:
                c.enc(PART_MODE, 0, 0)
                c.enc(PART_MODE, 1, 1 if p == 1 else 0)
                if p != 1 and log2 > 3:
                    c.enc(PART_MODE, 2, 1 if p == 2 else 0)
            return p
        if not self.amp:
            p = r.choice((0, 1, 2))
            c.enc(PART_MODE, 0, 1 if p == 0 else 0)
            if p:
                c.enc(PART_MODE, 1, 1 if p == 1 else 0)
            return p
        p = r.choice((0, 0, 1, 2, 4, 5, 6, 7))
        c.enc(PART_MODE, 0, 1 if p == 0 else 0)
        if p == 0:
            return 0
        horiz = p in (1, 4, 5)
        c.enc(PART_MODE, 1, 1 if horiz else 0)
        c.enc(PART_MODE, 3, 1 if p in (1, 2) else 0)
        if p not in (1, 2):
            c.byp(1 if p in (5, 7) else 0)
        return p

    def merge_idx(self):
        c, r = self.c, self.rng
        if self.max_merge > 1:
            i = r.randrange(self.max_merge)
            c.enc(MERGE_IDX, 0, 1 if i else 0)
            for k in range(1, self.max_merge - 1):
                if i >= k:
                    c.byp(1 if i > k else 0)

    def mvd(self):
        c, r = self.c, self.rng
        v = [int(r.gauss(0, 6)) if r.random() < 0.8 else r.randrange(-200, 201) for _ in range(2)]
        g0 = [int(t != 0) for t in v]
        g1 = [int(abs(t) > 1) for t in v]
        c.enc(MVD_G0, 0, g0[0]); c.enc(MVD_G0, 0, g0[1])
        if g0[0]:
            c.enc(MVD_G1, 1, g1[0])
        if g0[1]:
            c.enc(MVD_G1, 1, g1[1])
        for k in range(2):
            if not g0[k]:
                continue
            if g1[k]:
                rem, e = abs(v[k]) - 2, 1                           # EG1
                while rem >= (1 << e):
                    c.byp(1)
                    rem -= 1 << e
                    e += 1
                c.byp(0)
                c.byps(e, rem)
            c.byp(1 if v[k] < 0 else 0)

    def prediction_unit(self, pw, ph):
        """hls_prediction_unit (hevcdec.c:1717-1790): merge, or per list reference index + vector difference + predictor flag"""
        c, r = self.c, self.rng
        merge = int(r.random() < 0.4)
        c.enc(MERGE_FLAG, 0, merge)
        if merge:
            self.merge_idx()
            return 1
        idc = 0                                                      # PRED_L0 0, L1 1, BI 2
        if self.stype == 0:
            idc = r.choice((0, 1, 2)) if pw + ph != 12 else r.choice((0, 1))
            if pw + ph != 12:
                c.enc(INTER_IDC, self.ct_depth, 1 if idc == 2 else 0)
            if idc != 2:
                c.enc(INTER_IDC, 4, idc)
        for lst in (0, 1):
            if (lst == 0 and idc == 1) or (lst == 1 and idc == 0):
                continue
            if self.nact > 1:
                ri = r.randrange(self.nact)
                mx = self.nact - 1
                for k in range(min(mx, 2)):
                    c.enc(REF_L0, k, 1 if ri > k else 0)
                    if ri <= k:
                        break
                if ri >= 2:
                    for k in range(2, mx):
                        c.byp(1 if ri > k else 0)
                        if ri <= k:
                            break
            self.mvd()
            c.enc(MVP_FLAG, 0, r.randrange(2))
        return 0

    # ---- intra prediction unit (intra_prediction_unit + luma_intra_pred_mode)
    def candidates(self, x0, y0):
        lc = (1 << self.log2_ctb) - 1
        up = self.nb(self.ipm, x0, y0 - 1) if (y0 & lc) else 1
        left = self.nb(self.ipm, x0 - 1, y0) if (self.left_ok or (x0 & lc)) else 1
        if left == up:
            if left < 2:
                return [0, 1, 26]
            return [left, 2 + ((left - 2 - 1 + 32) & 31), 2 + ((left - 2 + 1) & 31)]
        cand = [left, up]
        cand.append(0 if 0 not in cand else 1 if 1 not in cand else 26)
        return cand

    def intra_pu(self, x0, y0, log2, nxn):
        c, r = self.c, self.rng
        side = 2 if nxn else 1
        pb = (1 << log2) >> (1 if nxn else 0)
        prev = [int(r.random() < 0.4) for _ in range(side * side)]
        for f in prev:
            c.enc(PREV_INTRA, 0, f)
        self.modes = []
        for i in range(side):
            for j in range(side):
                x, y = x0 + pb * j, y0 + pb * i
                cand = self.candidates(x, y)
                if prev[2 * i + j]:
                    k = r.randrange(3)
                    c.byp(1 if k else 0)
                    if k:
                        c.byp(1 if k > 1 else 0)
                    mode = cand[k]
                else:
                    rem = r.randrange(32)
                    c.byps(5, rem)
                    mode = rem
                    for cv in sorted(cand):
                        if mode >= cv:
                            mode += 1
                self.modes.append(mode)
                self.fill(self.ipm, x, y, pb, mode)
        cm = r.choice((4, 4, 0, 1, 2, 3))
        c.enc(CHROMA_MODE, 0, 0 if cm == 4 else 1)
        if cm != 4:
            c.byps(2, cm)
            t = (0, 26, 10, 1)[cm]
            self.mode_c = 34 if self.modes[0] == t else t
        else:
            self.mode_c = self.modes[0]

    # ---- transform tree / unit
    def transform_tree(self, x0, y0, xb, yb, log2, depth, blk, cbf_cb, cbf_cr):
        c, r = self.c, self.rng
        if self.intra_cu:
            if self.intra_split:
                if depth == 1:
                    self.cur_mode = self.modes[blk]
            else:
                self.cur_mode = self.modes[0]
        if log2 <= self.log2_max_tb and log2 > self.log2_min_tb and depth < self.max_depth and not (self.intra_split and depth == 0):
            split = int(r.random() < 0.4)
            c.enc(SPLIT_TT, 5 - log2, split)
        else:
            inter_split = self.depth_inter == 0 and not self.intra_cu and self.part != 0 and depth == 0
            split = int(log2 > self.log2_max_tb or (self.intra_split and depth == 0) or inter_split)
        if log2 > 2 and (depth == 0 or cbf_cb):
            cbf_cb = int(r.random() < 0.5)
            c.enc(CBF_C, depth, cbf_cb)
        elif log2 > 2 or depth == 0:
            cbf_cb = 0
        if log2 > 2 and (depth == 0 or cbf_cr):
            cbf_cr = int(r.random() < 0.5)
            c.enc(CBF_C, depth, cbf_cr)
        elif log2 > 2 or depth == 0:
            cbf_cr = 0
        if split:
            h = 1 << (log2 - 1)
            for k, (x, y) in enumerate(((x0, y0), (x0 + h, y0), (x0, y0 + h), (x0 + h, y0 + h))):
                self.transform_tree(x, y, x0, y0, log2 - 1, depth + 1, k, cbf_cb, cbf_cr)
            return
        cbf_luma = 1
        if self.intra_cu or depth != 0 or cbf_cb or cbf_cr:
            cbf_luma = int(r.random() < 0.6)
            c.enc(CBF_LUMA, 0 if depth else 1, cbf_luma)
        if not (cbf_luma or cbf_cb or cbf_cr):
            return
        if self.qp_delta and not self.qp_coded:
            d = r.choice((0, 0, 1, -1, 2, -3, 7, -9))
            a = abs(d)
            for k in range(min(a, 5)):
                c.enc(QP_DELTA, 1 if k else 0, 1)
            if a < 5:
                c.enc(QP_DELTA, 1 if a else 0, 0)
            else:
                rem, e = a - 5, 0                                    # EG0
                while rem >= (1 << e):
                    c.byp(1)
                    rem -= 1 << e
                    e += 1
                c.byp(0)
                c.byps(e, rem)
            if d:
                c.byp(1 if d < 0 else 0)
            self.qp_coded = 1
        scan = scan_c = SCAN_DIAG
        if self.intra_cu and log2 < 4:
            scan = SCAN_VERT if 6 <= self.cur_mode <= 14 else SCAN_HORIZ if 22 <= self.cur_mode <= 30 else SCAN_DIAG
            scan_c = SCAN_VERT if 6 <= self.mode_c <= 14 else SCAN_HORIZ if 22 <= self.mode_c <= 30 else SCAN_DIAG
        if cbf_luma:
            self.residual(log2, scan, 0)
        if log2 > 2:
            if cbf_cb:
                self.residual(log2 - 1, scan_c, 1)
            if cbf_cr:
                self.residual(log2 - 1, scan_c, 2)
        elif blk == 3:
            if cbf_cb:
                self.residual(log2, scan_c, 1)
            if cbf_cr:
                self.residual(log2, scan_c, 2)

    # ---- residual_coding (hls_residual_coding + hevc_cabac.c:718-873)
    def residual(self, log2, scan_idx, ci):
        c, r = self.c, self.rng
        size = 1 << log2
        if self.tskip and not self.bypass_cu and log2 == 2:
            c.enc(TSKIP, 1 if ci else 0, int(r.random() < 0.4))
        cgs, offs = scan_tables(log2, scan_idx)
        total = 16 * len(cgs)
        # the coefficients, in scan order: a last position, then sparse levels below it
        style = r.random()
        last = 0 if style < 0.15 else r.randrange(min(total, 16)) if style < 0.6 else r.randrange(total)
        lev = [0] * total
        for n in range(last + 1):
            if n == last or r.random() < self.density:
                a = 1 + min(int(r.expovariate(0.9)), 400) if r.random() < 0.93 else r.randrange(1, 2000)
                lev[n] = -a if r.randrange(2) else a
        xcg, ycg = cgs[last >> 4]
        xo, yo = offs[last & 15]
        lx, ly = 4 * xcg + xo, 4 * ycg + yo
        ex, ey = (ly, lx) if scan_idx == SCAN_VERT else (lx, ly)

        def prefix_of(v):
            if v < 4:
                return v, 0, 0
            p = 4
            while True:
                ln = (p >> 1) - 1
                base = (1 << ln) * (2 + (p & 1))
                if base <= v < base + (1 << ln):
                    return p, ln, v - base
                p += 1
        if ci == 0:
            ctx_off, ctx_sh = 3 * (log2 - 2) + ((log2 - 1) >> 2), (log2 + 1) >> 2
        else:
            ctx_off, ctx_sh = 15, log2 - 2
        mx = (log2 << 1) - 1
        px, py = prefix_of(ex), prefix_of(ey)
        for elem, (p, _ln, _s) in ((LAST_X, px), (LAST_Y, py)):
            for i in range(p):
                c.enc(elem, (i >> ctx_sh) + ctx_off, 1)
            if p < mx:
                c.enc(elem, (p >> ctx_sh) + ctx_off, 0)
        for (p, ln, s) in (px, py):
            if p > 3:
                c.byps(ln, s)

        ncg = 1 << (log2 - 2)
        cgflag = [[0] * 8 for _ in range(8)]
        last_subset = last >> 4
        g1ctx = 1
        for i in range(last_subset, -1, -1):
            x_cg, y_cg = cgs[i]
            sub = lev[16 * i:16 * i + 16]
            implicit = 0
            if 0 < i < last_subset:
                ctx_cg = 0
                if x_cg < ncg - 1:
                    ctx_cg += cgflag[x_cg + 1][y_cg]
                if y_cg < ncg - 1:
                    ctx_cg += cgflag[x_cg][y_cg + 1]
                cgflag[x_cg][y_cg] = int(any(sub))
                c.enc(SIG_CG, min(ctx_cg, 1) + (2 if ci else 0), cgflag[x_cg][y_cg])
                implicit = 1
            else:
                cgflag[x_cg][y_cg] = 1
            if i == last_subset:
                n_end = (last & 15) - 1
                sig_idx = [last & 15]
            else:
                n_end = 15
                sig_idx = []
            prev_sig = 0
            if x_cg < ncg - 1:
                prev_sig = cgflag[x_cg + 1][y_cg]
            if y_cg < ncg - 1:
                prev_sig += cgflag[x_cg][y_cg + 1] << 1
            for n in range(n_end, -1, -1):
                x_c, y_c = 4 * x_cg + offs[n][0], 4 * y_cg + offs[n][1]
                if cgflag[x_cg][y_cg] and (n > 0 or not implicit):
                    sig = int(sub[n] != 0)
                    c.enc(SIG_COEFF, self.sig_ctx(ci, x_c, y_c, log2, scan_idx, prev_sig), sig)
                    if sig:
                        sig_idx.append(n)
                        implicit = 0
                elif n == 0 and implicit and cgflag[x_cg][y_cg]:
                    assert sub[0] != 0
                    sig_idx.append(0)
            if not sig_idx:
                continue
            ctx_set = 2 if (i > 0 and ci == 0) else 0
            if i != last_subset and g1ctx == 0:
                ctx_set += 1
            g1ctx = 1
            first_g1 = -1
            g1 = {}
            for m, n in enumerate(sig_idx[:8]):
                f = int(abs(sub[n]) > 1)
                g1[n] = f
                c.enc(GT1, (ctx_set << 2) + g1ctx + (16 if ci else 0), f)
                if f:
                    g1ctx = 0
                elif 0 < g1ctx < 3:
                    g1ctx += 1
                if f and first_g1 == -1:
                    first_g1 = n
            g2 = 0
            if first_g1 != -1:
                g2 = int(abs(sub[first_g1]) > 2)
                c.enc(GT2, ctx_set + (4 if ci else 0), g2)
            for n in sig_idx:
                c.byp(1 if sub[n] < 0 else 0)
            rice = 0
            for m, n in enumerate(sig_idx):
                a = abs(sub[n])
                base = 1 + g1.get(n, 0) + (g2 if n == first_g1 else 0)
                if base == ((3 if n == first_g1 else 2) if m < 8 else 1):
                    self.remaining(a - base, rice)
                    if a > 3 * (1 << rice):
                        rice = min(rice + 1, 4)
                else:
                    assert a == base, (a, base, m)

    def remaining(self, v, k):
        c = self.c
        if v < (3 << k):
            for _ in range(v >> k):
                c.byp(1)
            c.byp(0)
            c.byps(k, v & ((1 << k) - 1))
            return
        w = v - (3 << k)
        p3 = 0
        while w >= (((1 << (p3 + 1)) - 1) << k):
            p3 += 1
        for _ in range(3 + p3):
            c.byp(1)
        c.byp(0)
        c.byps(p3 + k, w - (((1 << p3) - 1) << k))

    @staticmethod
    def sig_ctx(ci, x_c, y_c, log2, scan_idx, prev_sig):
        if x_c + y_c == 0:
            s = 0
        elif log2 == 2:
            s = (0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8)[(y_c << 2) + x_c]
        else:
            xo, yo = x_c & 3, y_c & 3
            if prev_sig == 0:
                s = 2 if xo + yo == 0 else 1 if xo + yo <= 2 else 0
            elif prev_sig == 1:
                s = 2 - min(yo, 2)
            elif prev_sig == 2:
                s = 2 - min(xo, 2)
            else:
                s = 2
            if ci == 0 and ((x_c >> 2) > 0 or (y_c >> 2) > 0):
                s += 3
            if log2 == 3:
                s += 9 if scan_idx == SCAN_DIAG else 15
            else:
                s += 12 if ci else 21
        return s if ci == 0 else s + 27

    def build(self):
        pkts = []
        for p in range(self.pictures):
            data = (self.vps() + self.sps() + self.pps() if p == 0 else b"") + self.picture(p)
            pkts.append(data)
        return pkts


STREAMS = {
    "i_8bit": dict(seed=1),
    "i_10bit": dict(seed=2, bd=10, w=80, h=72),
    "i_ctb16_slices": dict(seed=3, log2_ctb=4, log2_max_tb=4, slices=4, w=104, h=56, dbf_offsets=(2, -1)),
    "i_ctb64": dict(seed=4, log2_ctb=6, w=136, h=72, depth_intra=3, sao=2),
    "i_qpdelta_tskip": dict(seed=5, qp_delta=1, tskip=1, cb_off=3, cr_off=-4, qp=24),
    "i_bypass_nodbf": dict(seed=6, bypass=1, dbf_off=1, strong=0),
    "i_bypass_filtered": dict(seed=9, bypass=1, sao=2, qp=34),
    "i_pcm_unfiltered": dict(seed=10, pcm=1, pcm_lf_off=1, dbf_off=1, sao=0),
    "i_pcm_lf_off_10bit": dict(seed=15, pcm=1, pcm_lf_off=1, bd=10, sao=2, slices=2),
    # without SAO: what contrib/libav/mi355_hevc_lf_bridge.c deblocks per picture
    "i_nosao_8bit": dict(seed=21, sao=0, qp_delta=1, dbf_offsets=(1, -2), slices=3, pcm=1, pcm_lf_off=1, bypass=1, cb_off=2, cr_off=-3),
    "i_nosao_10bit_ctb64": dict(seed=22, sao=0, bd=10, log2_ctb=6, w=136, h=72, depth_intra=3, qp=36),
    "pb_nosao_8bit": dict(seed=23, sao=0, inter=1, pictures=5, qp_delta=1, qp=34),
    "pb_nosao_ctb16_10bit": dict(seed=24, sao=0, inter=1, pictures=4, bd=10, log2_ctb=4, log2_max_tb=4, slices=3, w=104, h=56, qp=38),
    "pb_pcm": dict(seed=16, pcm=1, inter=1, pictures=4, log2_ctb=4, log2_max_tb=4),
    "i_mincb16": dict(seed=7, log2_min_cb=4, log2_min_tb=3, depth_intra=1, w=96, h=96, qp=38, density=0.6),
    "i_scaling_10bit": dict(seed=8, scaling=1, bd=10, slices=3, across=0),
    "pb_8bit": dict(seed=11, inter=1, pictures=5),
    "pb_10bit_weighted": dict(seed=12, inter=1, pictures=5, bd=10, weighted=1, w=80, h=72),
    "pb_ctb16_slices_cip": dict(seed=13, inter=1, pictures=4, log2_ctb=4, log2_max_tb=4, slices=3, cip=1, w=104, h=56, amp=0),
    "pb_ctb64_depth0": dict(seed=14, inter=1, pictures=4, log2_ctb=6, w=136, h=72, depth_inter=0, depth_intra=1, weighted=1),
    # tiles (hevc_ps.c setup_pps, hls_decode_neighbour hevcdec.c:2255-2300, the tile rules of hevc_filter.c): uniform and explicit
    # grids, filtering across tile edges on and off, slices of whole tiles and tiles of whole slices
    "i_tiles_2x2": dict(seed=41, tiles=(2, 2), across_tiles=0, log2_ctb=4, log2_max_tb=4, w=104, h=72, sao=2, slices=2),
    "i_tiles_3x1_10bit": dict(seed=42, tiles=(3, 1), across_tiles=1, bd=10, w=136, h=72, sao=2, qp_delta=1, slices=2),
    "pb_tiles_2x3_noacross": dict(seed=43, tiles=(2, 3), across_tiles=0, across=0, inter=1, pictures=5, log2_ctb=4, log2_max_tb=4, w=104, h=88,
                                  sao=2, slices=3, tile_sizes=((5, 2), (1, 3, 2))),
    "pb_tiles_3x2_nosao_10bit": dict(seed=44, tiles=(3, 2), across_tiles=0, inter=1, pictures=4, bd=10, w=136, h=104, sao=0, slices=2, weighted=1,
                                     dbf_offsets=(1, -1)),
    # pictures that leave the decoder in another order than they enter it, references before and after (hevc_refs.c: bumping, frame RPS)
    "pb_pyramid": dict(seed=45, inter=1, pictures=10, pyramid=1, sao=2, w=112, h=80, weighted=1),
    "pb_pyramid_tiles_10bit": dict(seed=46, inter=1, pictures=7, pyramid=1, bd=10, tiles=(2, 2), across_tiles=0, w=136, h=104, slices=2),
    # wavefronts (entropy_coding_sync: contexts of a row from the second block of the row above) and dependent slice segments
    # (tab_slice_address stays the slice's: no slice edge between the segments)
    "i_wpp": dict(seed=47, wpp=1, slices=2, w=136, h=104, sao=2, across=0),
    "pb_wpp_dep_10bit": dict(seed=48, wpp=1, dep=1, slices=4, inter=1, pictures=4, bd=10, log2_ctb=4, log2_max_tb=4, w=104, h=88, sao=2, across=0),
    "pb_dep_slices": dict(seed=49, dep=1, slices=5, inter=1, pictures=4, w=136, h=104, sao=2, across=0, qp_delta=1),
    "pb_tiles_dep": dict(seed=50, dep=1, slices=3, tiles=(2, 2), across_tiles=0, across=0, inter=1, pictures=4, log2_ctb=4, log2_max_tb=4, w=104, h=88, sao=2),
    "pb_9bit": dict(seed=51, bd=9, inter=1, pictures=4, w=112, h=80, sao=2, weighted=1, pcm=1, tskip=1, qp_delta=1),
    # every slice with its own number of active references: the slices of a picture use different lists (the filter bridge must take the
    # decoder's boundary strengths for such pictures: the calls it had put aside are run then)
    "pb_slices_own_lists": dict(seed=52, inter=1, pictures=6, slices=4, vary_refs=1, sao=2, w=136, h=104, weighted=1, across=0),
    "pb_480p_ctb64": dict(seed=31, inter=1, pictures=4, log2_ctb=6, w=832, h=480, depth_inter=1, depth_intra=2, sao=2),
    "pb_1080p_ctb64": dict(seed=32, inter=1, pictures=5, log2_ctb=6, w=1920, h=1080, depth_inter=1, depth_intra=2, sao=2),
    "pb_1080p_few_intra": dict(seed=33, inter=1, pictures=6, log2_ctb=6, w=1920, h=1080, depth_inter=1, depth_intra=2, sao=2, intra_frac=0.02),
}


def write_samples(path, pkts):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0))
        f.write(struct.pack("<I", len(pkts)))
        for p in pkts:
            f.write(struct.pack("<I", len(p)))
            f.write(p)


def decode(path, exe, plain=True):
    os.makedirs(os.path.join(ROOT, "build", "streams"), exist_ok=True)
    out = os.path.join(ROOT, "build", "streams", os.path.basename(path) + (".plain.yuv" if plain else ".hook.yuv"))
    env = dict(os.environ)
    if plain:
        env["MI355_TIER1_PLAIN"] = "1"
    r = subprocess.run([exe, path, out], env=env, capture_output=True, text=True)
    data = open(out, "rb").read() if os.path.exists(out) else b""
    return r.returncode, r.stderr, data


def main():
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(ROOT, "oracle", "_ref", "hevc_tier1_emu")
    gold_path = os.path.join(HERE, "hevc_streams.json")
    gold = {}
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, kw in STREAMS.items():
        if only and name not in only:
            continue
        kw_obj = Hevc(name, **kw)
        pkts = kw_obj.build()
        path = os.path.join(OUT, "hevc_synth_%s.samples" % name)
        write_samples(path, pkts)
        rc, err, data = decode(path, exe)
        msgs = [l for l in err.splitlines() if not l.startswith("tier1:")]
        m = re.search(r"(\d+) packets, (\d+) pictures.* (\d+)x(\d+) (\w+),", err)
        # open loop: the decoder must have seen exactly the coding tree units and slice ends that were written
        cnt = re.search(r"(\d+) coding tree units in (\d+) slices", err)
        ok = rc == 0 and not msgs and m and int(m.group(2)) == kw.get("pictures", 2) and cnt and \
            (int(cnt.group(1)), int(cnt.group(2))) == (kw_obj.n_ctus, kw_obj.n_slices)
        print(name, "bytes", sum(map(len, pkts)), "->", err.strip().splitlines()[-1] if err.strip() else rc, "OK" if ok else "REJECTED")
        if not ok:
            print("\n".join(msgs[:8]))
            sys.exit(1)
        h = kw_obj
        if h.pcm and h.dbf_off and not h.sao:                        # nothing filters: the PCM samples must come out as written
            import numpy as np
            bps = 2 if h.bd > 8 else 1
            fsz = h.w * h.h * 3 // 2
            pic = np.frombuffer(data, np.uint16 if bps == 2 else np.uint8).reshape(-1, fsz)
            assert h.pcm_blocks
            for (poc, x0, y0, size, luma) in h.pcm_blocks:
                got = pic[poc][:h.w * h.h].reshape(h.h, h.w)[y0:y0 + size, x0:x0 + size]
                assert (got.reshape(-1) == np.array(luma) * 2).all(), (name, poc, x0, y0)
            print("   %d PCM blocks decode to the samples written" % len(h.pcm_blocks))
        gold[name] = {"md5": hashlib.md5(data).hexdigest(), "bytes": len(data), "pictures": int(m.group(2)), "width": int(m.group(3)), "height": int(m.group(4)),
                      "pix_fmt": m.group(5), "ctus": kw_obj.n_ctus, "slices": kw_obj.n_slices, "stream_md5": hashlib.md5(b"".join(pkts)).hexdigest()}
    if only:                                                         # named streams: their entries join the ones on file
        gold = dict(json.load(open(gold_path)), **gold)
    json.dump(gold, open(gold_path, "w"), indent=1, sort_keys=True)
    print("wrote", gold_path)


if __name__ == "__main__":
    main()
