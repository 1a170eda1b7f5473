#!/usr/bin/env python3
"""Generator of small H.264 test streams for the profiles the offline clips do not cover (High 4:2:2, High 10,
High 4:2:2 10 bit) and for 4:2:0 features they lack (several slices, loop filter off across slice edges, I_PCM, explicit
weights, several references, every partition shape).  TEST INFRASTRUCTURE: a bitstream WRITER (CAVLC) with random syntax
elements — no rate control, no motion search, nothing is "encoded": every element is drawn at random inside what the
syntax allows, and the streams' meaning is whatever the reference decoder makes of them.  The tests decode each stream
twice with the reference's own decoder (oracle/_ref/h264_tier1_*: DSP tables as the reference filled them / overridden
by this project's hooks) and compare the pictures; validity = the reference decoder decodes every picture without a
complaint (checked here when /root/reference's decoder harness exists, and again by the tests).

The CAVLC code tables are READ from the reference's source at generation time (libavcodec/h264_cavlc.c:48-238,
h264data.c:42-52) — the generated streams are committed (tests/golden/h264_synth_*.samples), this script needs
/root/reference only to regenerate them.

usage: make_h264_streams.py [outdir]        -> h264_synth_<name>.samples (the harness's format: u32 extradata_len = 0,
                                               u32 n, n x { u32 len, Annex-B access unit })"""
import os
import re
import struct
import sys

import numpy as np

REF = "/root/reference/libavcodec"


# ---------------------------------------------------------------- tables from the reference's source
def _table(src, name):
    m = re.search(r"\b%s\s*((?:\[[^\]]*\])+)\s*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", m.group(2), flags=re.S)
    vals = [int(v, 0) for v in re.findall(r"-?\b(?:0x[0-9a-fA-F]+|\d+)\b", body)]
    return vals


def load_tables():
    cav = open(os.path.join(REF, "h264_cavlc.c")).read()
    dat = open(os.path.join(REF, "h264data.c")).read()
    T = {}
    for n, shape in (("coeff_token_len", (4, 68)), ("coeff_token_bits", (4, 68)), ("chroma_dc_coeff_token_len", (20,)),
                     ("chroma_dc_coeff_token_bits", (20,)), ("chroma422_dc_coeff_token_len", (36,)), ("chroma422_dc_coeff_token_bits", (36,)),
                     ("total_zeros_len", (16, 16)), ("total_zeros_bits", (16, 16)), ("chroma_dc_total_zeros_len", (3, 4)),
                     ("chroma_dc_total_zeros_bits", (3, 4)), ("chroma422_dc_total_zeros_len", (7, 8)), ("chroma422_dc_total_zeros_bits", (7, 8)),
                     ("run_len", (7, 16)), ("run_bits", (7, 16))):
        v = _table(cav, n)
        a = np.zeros(int(np.prod(shape)), np.int64)
        a[:len(v)] = v            # rows the source leaves short are zero-filled, as in C
        # C fills row by row: re-read with row structure when rows are shorter than declared
        if len(shape) == 2 and len(v) != a.size:
            m = re.search(r"\b%s\s*(?:\[[^\]]*\])+\s*=\s*\{(.*?)\};" % n, cav, re.S)
            rows = re.findall(r"\{([^{}]*)\}", m.group(1))
            a = np.zeros(shape, np.int64)
            for r, row in enumerate(rows):
                rv = [int(x, 0) for x in re.findall(r"\b(?:0x[0-9a-fA-F]+|\d+)\b", re.sub(r"/\*.*?\*/", "", row, flags=re.S))]
                a[r, :len(rv)] = rv
            T[n] = a
        else:
            T[n] = a.reshape(shape)
    T["intra_cbp"] = _table(dat, "ff_h264_golomb_to_intra4x4_cbp")
    T["inter_cbp"] = _table(dat, "ff_h264_golomb_to_inter_cbp")
    assert len(T["intra_cbp"]) == 48 and sorted(T["intra_cbp"]) == list(range(48))
    T["intra_cbp_gray_code"] = {c: i for i, c in enumerate(_table(cav, "golomb_to_intra4x4_cbp_gray"))}
    T["inter_cbp_gray_code"] = {c: i for i, c in enumerate(_table(cav, "golomb_to_inter_cbp_gray"))}
    assert len(T["intra_cbp_gray_code"]) == 16 and len(T["inter_cbp_gray_code"]) == 16
    T["intra_cbp_code"] = {c: i for i, c in enumerate(T["intra_cbp"])}
    T["inter_cbp_code"] = {c: i for i, c in enumerate(T["inter_cbp"])}
    return T


# ---------------------------------------------------------------- bit writer
class Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        assert 0 <= v < (1 << n), (n, v)
        for i in range(n - 1, -1, -1):
            self.b.append((v >> i) & 1)

    def ue(self, v):
        assert v >= 0
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def te(self, rng, v):
        if rng > 1:
            self.ue(v)
        else:
            self.u(1, 1 - v)

    def vlc(self, length, bits):
        assert length > 0, "code not in the table"
        self.u(int(length), int(bits))

    def aligned(self):
        return len(self.b) % 8 == 0

    def align_zero(self):
        while len(self.b) % 8:
            self.b.append(0)

    def trailing(self):
        self.b.append(1)
        self.align_zero()

    def bytes(self):
        assert len(self.b) % 8 == 0
        return bytes(int("".join(map(str, self.b[i:i + 8])), 2) for i in range(0, len(self.b), 8))


def nal(ref_idc, typ, rbsp):
    out = bytearray(b"\x00\x00\x00\x01")
    out.append((ref_idc << 5) | typ)
    zeros = 0
    for x in rbsp:
        if zeros >= 2 and x <= 3:
            out.append(3)
            zeros = 0
        out.append(x)
        zeros = zeros + 1 if x == 0 else 0
    return bytes(out)


# ---------------------------------------------------------------- residual blocks (CAVLC, 9.2)
def write_block(w, T, coefs, nC, kind):
    """coefs: the block in scan order (len 16, 15, 4 or 8); kind: 'luma' (tables by nC), 'cdc420', 'cdc422'.  Returns total_coeff."""
    maxc = len(coefs)
    nz = [i for i, c in enumerate(coefs) if c]
    total = len(nz)
    lev = [coefs[i] for i in reversed(nz)]            # highest frequency first
    t1 = 0
    while t1 < min(3, total) and abs(lev[t1]) == 1:
        t1 += 1
    idx = 4 * total + t1
    if kind == "cdc420":
        w.vlc(T["chroma_dc_coeff_token_len"][idx], T["chroma_dc_coeff_token_bits"][idx])
    elif kind == "cdc422":
        w.vlc(T["chroma422_dc_coeff_token_len"][idx], T["chroma422_dc_coeff_token_bits"][idx])
    else:
        tab = 0 if nC < 2 else (1 if nC < 4 else (2 if nC < 8 else 3))
        w.vlc(T["coeff_token_len"][tab][idx], T["coeff_token_bits"][tab][idx])
    if total == 0:
        return 0
    for k in range(t1):
        w.u(1, 1 if lev[k] < 0 else 0)
    sl = 1 if (total > 10 and t1 < 3) else 0
    for k in range(t1, total):
        v = lev[k]
        code = 2 * abs(v) - 2 if v > 0 else 2 * abs(v) - 1
        if k == t1 and t1 < 3:
            code -= 2
        if sl == 0:
            if code < 14:
                w.u(code + 1, 1)
            elif code < 30:
                w.u(15, 1)
                w.u(4, code - 14)
            else:
                assert code - 30 < 4096
                w.u(16, 1)
                w.u(12, code - 30)
        else:
            if code < (15 << sl):
                w.u((code >> sl) + 1, 1)
                w.u(sl, code & ((1 << sl) - 1))
            else:
                assert code - (15 << sl) < 4096
                w.u(16, 1)
                w.u(12, code - (15 << sl))
        if sl == 0:
            sl = 1
        if abs(v) > (3 << (sl - 1)) and sl < 6:
            sl += 1
    if total < maxc:
        tz = nz[-1] + 1 - total
        if kind == "cdc420":
            w.vlc(T["chroma_dc_total_zeros_len"][total - 1][tz], T["chroma_dc_total_zeros_bits"][total - 1][tz])
        elif kind == "cdc422":
            w.vlc(T["chroma422_dc_total_zeros_len"][total - 1][tz], T["chroma422_dc_total_zeros_bits"][total - 1][tz])
        else:
            w.vlc(T["total_zeros_len"][total - 1][tz], T["total_zeros_bits"][total - 1][tz])
        left = tz
        pos = list(reversed(nz))
        for k in range(total - 1):
            if left <= 0:
                break
            run = pos[k] - pos[k + 1] - 1
            r = min(left, 7) - 1
            w.vlc(T["run_len"][r][run], T["run_bits"][r][run])
            left -= run
    return total


class Rng:
    def __init__(self, seed):
        self.r = np.random.default_rng(seed)

    def i(self, lo, hi):
        return int(self.r.integers(lo, hi + 1))

    def p(self, prob):
        return bool(self.r.random() < prob)

    def block(self, n, density, big=0.05):
        """n coefficients in scan order: mostly small, low frequencies more often"""
        out = [0] * n
        if not self.p(density):
            return out
        k = self.i(1, n)
        for j in range(k):
            if self.p(0.7 if j < 4 else 0.35):
                mag = self.i(1, 40) if self.p(big) else (1 if self.p(0.6) else self.i(2, 6))
                out[j] = mag if self.p(0.5) else -mag
        return out


# ---------------------------------------------------------------- a stream
class Stream:
    def __init__(self, T, name, mb_w, mb_h, chroma_idc, depth, seed, nslices=1, deblock_idc=0, weighted=True, nrefs=2, npics=6, far=9, bmode=0, t8x8=False, lossless=False, crop=None, mixed=False, cip=False, paff=False, sparse=1.0, skip=0.15, reorder=False, npps=1, scaling=False, gaps=False, mmco=False):
        self.T, self.name, self.mb_w, self.mb_h, self.cidc, self.depth = T, name, mb_w, mb_h, chroma_idc, depth
        self.r = Rng(seed)
        self.nslices, self.deblock_idc, self.weighted, self.nrefs, self.npics, self.far = nslices, deblock_idc, weighted, nrefs, npics, far
        self.t8x8 = t8x8                                     # transform_8x8_mode_flag: Intra 8x8 and the 8x8 transform of inter macroblocks
        self.bmode = bmode                                   # B pictures: 0 none, 1 implicit weights, 2 explicit weights, 3 plain average
        self.cblk_h = 4 if chroma_idc == 2 else 2            # chroma 4x4 blocks per macroblock, vertically
        self.qp_min, self.qp_max = 12, 44
        self.crop, self.mixed, self.cip = crop, mixed, cip   # (right, bottom) cropping in chroma-sample units; I and P slices in one picture; constrained_intra_pred
        self.sparse, self.skip = sparse, skip                # scale of the coded-block probabilities, P(skip): 1.0 / 0.15 = dense test content
        self.npps, self.scaling = npps, scaling              # picture parameter sets (chroma QP offsets differ; each slice picks one); scaling lists in them
        self.mmco = mmco                                     # adaptive reference marking: a long-term frame (picture 2), memory_management_control_operation 5 (picture 6)
        self.gaps = gaps                                     # gaps_in_frame_num_value_allowed_flag, and some frame_num values are skipped
        self.reorder = reorder                               # reference list modification in every P / B slice (the same picture may appear twice)
        self.paff = paff                                     # frame_mbs_only_flag 0: each frame is coded as a frame picture or as two field pictures
        self.lossless = lossless                             # qpprime_y_zero_transform_bypass_flag and QP'Y = 0 throughout: transform bypass
        if lossless:
            self.qp_min = self.qp_max = -6 * (depth - 8)

    def sps(self):
        w = Bits()
        profile = 244 if self.cidc == 3 else (122 if self.cidc == 2 else (110 if self.depth > 8 else 100))
        w.u(8, profile); w.u(8, 0); w.u(8, 40)
        w.ue(0)
        w.ue(self.cidc)
        if self.cidc == 3:
            w.u(1, 0)                 # separate_colour_plane_flag
        w.ue(self.depth - 8); w.ue(self.depth - 8); w.u(1, 1 if self.lossless else 0); w.u(1, 0)
        w.ue(0)                       # log2_max_frame_num - 4
        if self.bmode:
            w.ue(0); w.ue(2)          # pic_order_cnt_type 0, 6 bits of pic_order_cnt_lsb
        else:
            w.ue(2)                   # pic_order_cnt_type 2: output order = decoding order
        w.ue(max(1, self.nrefs)); w.u(1, 1 if self.gaps else 0)
        if self.paff:
            w.ue(self.mb_w - 1); w.ue(self.mb_h // 2 - 1)       # map units: field macroblock rows
            w.u(1, 0); w.u(1, 0); w.u(1, 1)                      # frame_mbs_only 0, mb_adaptive_frame_field 0, direct_8x8_inference
        else:
            w.ue(self.mb_w - 1); w.ue(self.mb_h - 1)
            w.u(1, 1); w.u(1, 1)
        if self.crop:
            w.u(1, 1); w.ue(0); w.ue(self.crop[0]); w.ue(0); w.ue(self.crop[1])      # left, right, top, bottom
        else:
            w.u(1, 0)
        if self.bmode:
            # VUI with nothing but the bitstream restrictions: one picture of reordering (the decoder need not guess)
            w.u(1, 1)
            for _ in range(4):
                w.u(1, 0)             # aspect ratio, overscan, video signal type, chroma location
            for _ in range(4):
                w.u(1, 0)             # timing, NAL HRD, VCL HRD, pic_struct
            w.u(1, 1)
            w.u(1, 1); w.ue(0); w.ue(0); w.ue(16); w.ue(16); w.ue(1); w.ue(max(2, self.nrefs))
        else:
            w.u(1, 0)
        w.trailing()
        return nal(3, 7, w.bytes())

    def pps(self, k=0):
        w = Bits()
        w.ue(k); w.ue(0); w.u(1, 0); w.u(1, 0); w.ue(0)
        w.ue(max(1, self.nrefs) - 1); w.ue(0)
        w.u(1, 1 if self.weighted else 0); w.u(2, (0, 2, 1, 0)[self.bmode])
        w.se(0); w.se(0); w.se((2, -4, 6, -9)[k & 3])
        w.u(1, 1); w.u(1, 1 if self.cip else 0); w.u(1, 0)
        w.u(1, 1 if self.t8x8 else 0)                        # transform_8x8_mode
        if self.scaling:
            # pic_scaling_matrix_present: some lists given (random entries 4..40), the others fall back (7.4.2.2 rules A / B)
            r = Rng(1000 + 17 * k + self.mb_w)
            w.u(1, 1)
            nlists = 6 + ((6 if self.cidc == 3 else 2) if self.t8x8 else 0)
            for i in range(nlists):
                present = r.p(0.6)
                w.u(1, int(present))
                if present:
                    last = 8
                    for _ in range(16 if i < 6 else 64):
                        nxt = r.i(4, 40)
                        w.se(nxt - last)
                        last = nxt
        else:
            w.u(1, 0)
        w.se((-3, 5, 0, -7)[k & 3])                          # second chroma qp offset
        w.trailing()
        return nal(3, 8, w.bytes())

    def param_sets(self):
        return self.sps() + b"".join(self.pps(k) for k in range(self.npps))

    # ---- neighbour bookkeeping of one picture
    def begin_picture(self):
        self.pic_pps = self.r.i(0, self.npps - 1) if self.npps > 1 else 0
        W4, H4 = 4 * self.mb_w, 4 * self.mb_h
        self.nnz = np.zeros((H4, W4), np.int64)
        self.nnz444 = [self.nnz, np.zeros((H4, W4), np.int64), np.zeros((H4, W4), np.int64)]      # 4:4:4: Cb and Cr coded like luma
        self.nnzc = np.zeros((2, self.cblk_h * self.mb_h, 2 * self.mb_w), np.int64)
        self.i4 = np.full((H4, W4), -1, np.int64)            # Intra4x4PredMode per block, -1: none
        self.kind = [[None] * self.mb_w for _ in range(self.mb_h)]
        self.slice_of = np.full((self.mb_h, self.mb_w), -1, np.int64)

    def avail(self, mbx, mby, sid):
        return 0 <= mbx < self.mb_w and 0 <= mby < self.mb_h and self.slice_of[mby, mbx] == sid

    def iavail(self, mbx, mby, sid):
        """available for INTRA prediction: with constrained_intra_pred an inter neighbour is not (8.3.1.2, 8.3.3, 8.3.4)"""
        if not self.avail(mbx, mby, sid):
            return False
        return not self.cip or self.kind[mby][mbx] in ("i4", "i8", "i16", "pcm")

    def nbrs(self, x, y, bw, bh, mbx, mby, sid):
        """storage positions (row, column) of the blocks to the left of and above block (x, y) of a grid with bw x bh blocks per
        macroblock, or None where that neighbour is not available (6.4.11.4; MbaffStream: 6.4.12.2)"""
        a = (y, x - 1) if (x % bw or self.avail(mbx - 1, mby, sid)) and x > 0 else None
        b = (y - 1, x) if (y % bh or self.avail(mbx, mby - 1, sid)) and y > 0 else None
        return a, b

    def intra_avail(self, mbx, mby, sid):
        """left, top, top-left macroblock usable for intra prediction"""
        return self.iavail(mbx - 1, mby, sid), self.iavail(mbx, mby - 1, sid), self.iavail(mbx - 1, mby - 1, sid)

    def left_rows(self, mbx, mby, sid, left):
        """per row of 4x4 blocks: the samples to the left usable for intra prediction; the block to the left counts in the derivation of
        the predicted Intra4x4PredMode (MbaffStream: the two differ, and differ by row)"""
        return [left] * 4, [left] * 4

    def ref_range(self, nact):
        return nact

    def nC(self, arr, x, y, bw, bh, mbx, mby, sid):
        """predicted count for the block at block coordinates (x, y) of an array with bw x bh blocks per macroblock"""
        pa, pb = self.nbrs(x, y, bw, bh, mbx, mby, sid)
        a = arr[pa] if pa is not None else None
        b = arr[pb] if pb is not None else None
        if a is not None and b is not None:
            return (int(a) + int(b) + 1) >> 1
        return int(a) if a is not None else (int(b) if b is not None else 0)

    # ---- macroblocks
    def residual(self, w, mbx, mby, sid, cbp, i16):
        T, r = self.T, self.r
        for pl in range(3 if self.cidc == 3 else 1):
            nnz = self.nnz444[pl]
            if i16:
                n = self.nC(nnz, 4 * mbx, 4 * mby, 4, 4, mbx, mby, sid)
                write_block(w, T, r.block(16, 0.8 * self.sparse), n, "luma")
            for blk in range(16):
                x = 4 * mbx + (blk & 1) + 2 * ((blk >> 2) & 1)
                y = 4 * mby + ((blk >> 1) & 1) + 2 * (blk >> 3)
                if cbp & (1 << (blk >> 2)):
                    n = self.nC(nnz, x, y, 4, 4, mbx, mby, sid)
                    nnz[y, x] = write_block(w, T, r.block(15 if i16 else 16, 0.6 * self.sparse), n, "luma")
                else:
                    nnz[y, x] = 0
        if self.cidc == 3:
            return
        cc = cbp >> 4
        nblk = 2 * self.cblk_h
        if cc:
            for _ in range(2):
                write_block(w, T, r.block(nblk, 0.7 * self.sparse), 0, "cdc422" if self.cidc == 2 else "cdc420")
        for pl in range(2):
            for blk in range(nblk):
                # 4:2:0: raster order of the 2 x 2 blocks; 4:2:2: two 2 x 2 groups, top then bottom (blkIdx 0..7)
                x = 2 * mbx + (blk & 1)
                y = self.cblk_h * mby + (blk >> 1)
                if cc & 2:
                    n = self.nC(self.nnzc[pl], x, y, 2, self.cblk_h, mbx, mby, sid)
                    self.nnzc[pl][y, x] = write_block(w, T, r.block(15, 0.5 * self.sparse), n, "luma")
                else:
                    self.nnzc[pl][y, x] = 0

    def intra_tail(self, w, cmode):
        """intra_chroma_pred_mode and coded_block_pattern of an Intra NxN macroblock (4:4:4: no chroma mode, luma bits only)"""
        r = self.r
        cbp = r.i(0, 15) | (r.i(0, 2) << 4)
        if self.cidc == 3:
            cbp &= 15
            w.ue(self.T["intra_cbp_gray_code"][cbp])
        else:
            w.ue(cmode)
            w.ue(self.T["intra_cbp_code"][cbp])
        return cbp

    def inter_cbp(self, w, prob):
        r = self.r
        cbp = (r.i(0, 15) | (r.i(0, 2) << 4)) if r.p(prob) else 0
        if self.cidc == 3:
            cbp &= 15
            w.ue(self.T["inter_cbp_gray_code"][cbp])
        else:
            w.ue(self.T["inter_cbp_code"][cbp])
        return cbp

    def clear_counts(self, mbx, mby, value=0):
        for a in self.nnz444:
            a[4 * mby:4 * mby + 4, 4 * mbx:4 * mbx + 4] = value
        self.nnzc[:, self.cblk_h * mby:self.cblk_h * (mby + 1), 2 * mbx:2 * mbx + 2] = value

    def qp_delta(self, w):
        d = self.r.i(-3, 3) if self.r.p(0.5) else 0
        if not self.qp_min <= self.qp + d <= self.qp_max:
            d = 0
        self.qp += d
        w.se(d)

    def intra_mb(self, w, mbx, mby, sid, base):
        """an intra macroblock; base: mb_type offset of intra types in this slice type (0 in I, 5 in P)"""
        r = self.r
        left, top, topleft = self.intra_avail(mbx, mby, sid)
        ls, lm = self.left_rows(mbx, mby, sid, left)
        c = r.i(0, 9)
        if c == 0:                                           # I_PCM
            w.ue(base + 25)
            w.align_zero()
            n = 768 if self.cidc == 3 else 256 + 2 * 8 * (16 if self.cidc == 2 else 8)
            for _ in range(n):
                w.u(self.depth, r.i(0, (1 << self.depth) - 1))
            self.clear_counts(mbx, mby, 16)
            self.kind[mby][mbx] = "pcm"
            return
        cmodes = [0] + ([1] if left else []) + ([2] if top else []) + ([3] if left and top and topleft else [])
        cmode = cmodes[r.i(0, len(cmodes) - 1)]
        if c <= 4 and self.t8x8 and r.p(0.5):                # Intra 8x8
            w.ue(base + 0)
            w.u(1, 1)
            for b8 in range(4):
                bx, by = 2 * (b8 & 1), 2 * (b8 >> 1)
                x, y = 4 * mbx + bx, 4 * mby + by
                l_ok, t_ok = bx > 0 or (ls[by] and ls[by + 1]), by > 0 or top
                tl_ok = True if (bx > 0 and by > 0) else (top if bx > 0 else ((ls[by - 1] and ls[by]) if by > 0 else topleft))
                ok = [2] + ([0, 3, 7] if t_ok else []) + ([1, 8] if l_ok else []) + ([4, 5, 6] if l_ok and t_ok and tl_ok else [])
                mode = ok[r.i(0, len(ok) - 1)]
                pa, pb = self.nbrs(x, y, 4, 4, mbx, mby, sid)
                if pa is not None and not (bx > 0 or lm[by]):
                    pa = None
                if pb is not None and not (by > 0 or top):
                    pb = None
                ma = None if pa is None else (2 if self.i4[pa] < 0 else int(self.i4[pa]))
                mb_ = None if pb is None else (2 if self.i4[pb] < 0 else int(self.i4[pb]))
                pred = 2 if ma is None or mb_ is None else min(ma, mb_)
                if mode == pred:
                    w.u(1, 1)
                else:
                    w.u(1, 0)
                    w.u(3, mode if mode < pred else mode - 1)
                self.i4[y:y + 2, x:x + 2] = mode
            cbp = self.intra_tail(w, cmode)
            if cbp:
                self.qp_delta(w)
            self.residual(w, mbx, mby, sid, cbp, False)
            self.kind[mby][mbx] = "i8"
            return
        if c <= 4:                                           # Intra 4x4
            w.ue(base + 0)
            if self.t8x8:
                w.u(1, 0)
            for blk in range(16):
                bx, by = (blk & 1) + 2 * ((blk >> 2) & 1), ((blk >> 1) & 1) + 2 * (blk >> 3)
                x, y = 4 * mbx + bx, 4 * mby + by
                l_ok, t_ok = bx > 0 or ls[by], by > 0 or top
                # the sample above-left of the block lies in this macroblock, the one above, the one to the left or the one above-left
                tl_ok = True if (bx > 0 and by > 0) else (top if bx > 0 else ((ls[by - 1] and ls[by]) if by > 0 else topleft))
                ok = [2] + ([0, 3, 7] if t_ok else []) + ([1, 8] if l_ok else []) + ([4, 5, 6] if l_ok and t_ok and tl_ok else [])
                mode = ok[r.i(0, len(ok) - 1)]
                # predicted mode: min of the neighbours' modes; a neighbour outside -> 2 for both; an available neighbour that is
                # not Intra4x4 counts as 2 (8.3.1.1)
                pa, pb = self.nbrs(x, y, 4, 4, mbx, mby, sid)
                if pa is not None and not (bx > 0 or lm[by]):
                    pa = None                                # constrained intra: an inter neighbour does not count
                if pb is not None and not (by > 0 or top):
                    pb = None
                ma = None if pa is None else (2 if self.i4[pa] < 0 else int(self.i4[pa]))
                mb_ = None if pb is None else (2 if self.i4[pb] < 0 else int(self.i4[pb]))
                pred = 2 if ma is None or mb_ is None else min(ma, mb_)
                if mode == pred:
                    w.u(1, 1)
                else:
                    w.u(1, 0)
                    w.u(3, mode if mode < pred else mode - 1)
                self.i4[y, x] = mode
            cbp = self.intra_tail(w, cmode)
            if cbp:
                self.qp_delta(w)
            self.residual(w, mbx, mby, sid, cbp, False)
            self.kind[mby][mbx] = "i4"
            return
        modes = [2] + ([0] if top else []) + ([1] if left else []) + ([3] if left and top and topleft else [])
        mode = modes[r.i(0, len(modes) - 1)]
        cl, cc = r.i(0, 1), r.i(0, 2)
        if self.cidc == 3:
            cc = 0
        w.ue(base + 1 + mode + 4 * cc + 12 * cl)
        if self.cidc != 3:
            w.ue(cmode)
        self.qp_delta(w)
        self.residual(w, mbx, mby, sid, (15 if cl else 0) | (cc << 4), True)
        self.kind[mby][mbx] = "i16"

    def mvd(self, w):
        r = self.r
        for _ in range(2):
            w.se(r.i(-self.far, self.far) if r.p(0.7) else 0)

    def inter_mb(self, w, mbx, mby, sid, nact):
        r = self.r
        t = r.i(0, 4) if nact > 1 else r.i(0, 3)
        multi = self.ref_range(nact) > 1
        w.ue(t)
        small = False
        if t == 3 or t == 4:
            subs = [r.i(0, 3) for _ in range(4)]
            small = any(subs)
            for s_ in subs:
                w.ue(s_)
            if t == 3 and multi:
                for _ in range(4):
                    w.te(self.ref_range(nact) - 1, r.i(0, self.ref_range(nact) - 1))
            for s_ in subs:
                for _ in range((1, 2, 2, 4)[s_]):
                    self.mvd(w)
        else:
            parts = 1 if t == 0 else 2
            if multi:
                for _ in range(parts):
                    w.te(self.ref_range(nact) - 1, r.i(0, self.ref_range(nact) - 1))
            for _ in range(parts):
                self.mvd(w)
        cbp = self.inter_cbp(w, 0.7 * self.sparse)
        if self.t8x8 and (cbp & 15) and not small:
            w.u(1, r.i(0, 1))                                # transform_size_8x8_flag
        if cbp:
            self.qp_delta(w)
        self.residual(w, mbx, mby, sid, cbp, False)
        self.kind[mby][mbx] = "inter"

    def b_mb(self, w, mbx, mby, sid, nact):
        """a B macroblock: direct, 16x16 / 16x8 / 8x16 from list 0, list 1 or both, or four sub-macroblocks (Table 7-14, 7-18)"""
        r = self.r
        L0, L1, BI = 1, 2, 3
        t = r.i(0, 22)
        w.ue(t)
        if t == 22:
            subs = [r.i(0, 12) for _ in range(4)]
            for s_ in subs:
                w.ue(s_)
            spred = {0: 0, 1: L0, 2: L1, 3: BI, 4: L0, 5: L0, 6: L1, 7: L1, 8: BI, 9: BI, 10: L0, 11: L1, 12: BI}
            sparts = {0: 0, 1: 1, 2: 1, 3: 1, 4: 2, 5: 2, 6: 2, 7: 2, 8: 2, 9: 2, 10: 4, 11: 4, 12: 4}
            rr = self.ref_range(nact)
            for lst in (L0, L1):
                if rr > 1:
                    for s_ in subs:
                        if spred[s_] & lst:
                            w.te(rr - 1, r.i(0, rr - 1))
            for lst in (L0, L1):
                for s_ in subs:
                    if spred[s_] & lst:
                        for _ in range(sparts[s_]):
                            self.mvd(w)
        elif t > 0:
            if t <= 3:
                preds = [(L0, L1, BI)[t - 1]]
            else:
                pair = {4: (L0, L0), 6: (L1, L1), 8: (L0, L1), 10: (L1, L0), 12: (L0, BI), 14: (L1, BI), 16: (BI, L0), 18: (BI, L1), 20: (BI, BI)}
                preds = list(pair[t & ~1])
            rr = self.ref_range(nact)
            for lst in (L0, L1):
                if rr > 1:
                    for pr in preds:
                        if pr & lst:
                            w.te(rr - 1, r.i(0, rr - 1))
            for lst in (L0, L1):
                for pr in preds:
                    if pr & lst:
                        self.mvd(w)
        cbp = self.inter_cbp(w, 0.6 * self.sparse)
        if self.t8x8 and (cbp & 15) and not (t == 22 and any(s_ > 3 for s_ in subs)):
            w.u(1, r.i(0, 1))                                # transform_size_8x8_flag (direct_8x8_inference_flag = 1)
        if cbp:
            self.qp_delta(w)
        self.residual(w, mbx, mby, sid, cbp, False)
        self.kind[mby][mbx] = "inter"

    # ---- slices and pictures
    def list_modification(self, w, frame_num, held):
        """ref_pic_list_modification of one list: 1..held operations, each moving the short-term frame k frames back (k = 1..held,
        repeats allowed) to the next index — idc 0 (subtract) only, abs_diff_pic_num_minus1 relative to the running prediction (8.2.4.3.1)"""
        r = self.r
        if not (self.reorder and held > 0 and r.p(0.8)):
            w.u(1, 0)
            return
        w.u(1, 1)
        pred = frame_num & 15
        for _ in range(r.i(1, held)):
            target = (frame_num - r.i(1, held)) & 15
            w.ue(0)
            w.ue((pred - target - 1) & 15)
            pred = target
        w.ue(3)

    def weight_table(self, w, n, lists):
        r = self.r
        w.ue(r.i(2, 6)); w.ue(r.i(2, 6))
        for _ in range(lists):
            for _ in range(n):
                f = r.p(0.7)
                w.u(1, int(f))
                if f:
                    w.se(r.i(-20, 90)); w.se(r.i(-12, 12))
                f = r.p(0.7)
                w.u(1, int(f))
                if f:
                    for _ in range(2):
                        w.se(r.i(-20, 90)); w.se(r.i(-12, 12))

    def slice(self, idx, frame_num, idr, is_p, first_mb, last_mb, sid, nact, is_b=False, poc=None, ref_idc=3, field=None, marking=None):
        r = self.r
        w = Bits()
        w.ue(first_mb)
        w.ue(6 if is_b else (5 if is_p else 7))
        w.ue(self.pic_pps)                                   # every slice of a picture names the same set (the reference insists)
        w.u(4, frame_num & 15)
        if self.paff:
            w.u(1, 0 if field is None else 1)
            if field is not None:
                w.u(1, field)                                # bottom_field_flag
        if idr:
            w.ue(idx & 3)
        if poc is not None:
            w.u(6, poc & 63)
        if is_b:
            w.u(1, r.i(0, 1))                                # direct_spatial_mv_pred_flag
            w.u(1, 1)
            w.ue(nact - 1); w.ue(nact - 1)
            if self.reorder:
                self.list_modification(w, frame_num, nact); self.list_modification(w, frame_num, nact)
            else:
                w.u(1, 0); w.u(1, 0)                         # no reference list modification, either list
            if self.bmode == 2:
                self.weight_table(w, nact, 2)
        if is_p:
            w.u(1, 1)
            w.ue(nact - 1)
            if self.reorder:
                self.list_modification(w, frame_num, nact)
            else:
                w.u(1, 0)                                    # no reference list modification
            if self.weighted:
                w.ue(r.i(2, 6)); w.ue(r.i(2, 6))
                for _ in range(nact):
                    f = r.p(0.7)
                    w.u(1, int(f))
                    if f:
                        w.se(r.i(-20, 90)); w.se(r.i(-12, 12))
                    f = r.p(0.7)
                    w.u(1, int(f))
                    if f:
                        for _ in range(2):
                            w.se(r.i(-20, 90)); w.se(r.i(-12, 12))
        if idr:
            w.u(1, 0); w.u(1, 0)
        elif ref_idc and marking == "long":
            w.u(1, 1)
            w.ue(4); w.ue(1)                                 # one long-term frame index
            w.ue(3); w.ue(0); w.ue(0)                        # the previous frame becomes long-term frame 0
            w.ue(0)
        elif ref_idc and marking == "reset":
            w.u(1, 1); w.ue(5); w.ue(0)                      # all reference pictures unused; this picture becomes frame_num 0
        elif ref_idc:
            w.u(1, 0)
        self.qp = 26 + (0 if idx == 0 else r.i(-6, 6))
        if self.lossless:
            self.qp = self.qp_min
        w.se(self.qp - 26)
        idc = self.deblock_idc if self.deblock_idc >= 0 else r.i(0, 2)     # -1: every slice draws its own
        w.ue(idc)
        if idc != 1:
            w.se(r.i(-2, 2)); w.se(r.i(-2, 2))
        skip = 0
        for a in range(first_mb, last_mb):
            mbx, mby = a % self.mb_w, a // self.mb_w
            self.slice_of[mby, mbx] = sid
            if is_p or is_b:
                if r.p(self.skip):
                    skip += 1
                    self.clear_counts(mbx, mby)
                    self.kind[mby][mbx] = "skip"
                    continue
                w.ue(skip)
                skip = 0
                if r.p((0.25 if is_p else 0.15) * min(1.0, self.sparse * 2)):
                    self.intra_mb(w, mbx, mby, sid, 5 if is_p else 23)
                elif is_b:
                    self.b_mb(w, mbx, mby, sid, nact)
                else:
                    self.inter_mb(w, mbx, mby, sid, nact)
            else:
                self.intra_mb(w, mbx, mby, sid, 0)
        if (is_p or is_b) and skip:
            w.ue(skip)
        w.trailing()
        return nal(ref_idc, 5 if idr else 1, w.bytes())

    def build_paff(self):
        """an IDR frame picture, then every frame as a frame picture or as a top and a bottom field picture (I or P)"""
        r = self.r
        units, frame_h, nfr = [], self.mb_h, 0
        assert frame_h % 2 == 0 and not self.bmode
        for i in range(self.npics):
            idr = i == 0
            au = self.param_sets() if idr else b""
            fields = (None,) if (idr or r.p(0.4)) else (0, 1)
            for fld in fields:
                self.mb_h = frame_h if fld is None else frame_h // 2
                nmb = self.mb_w * self.mb_h
                self.begin_picture()
                held = min(nfr, max(1, self.nrefs))
                avail = held if fld is None else 2 * held + (1 if fld == 1 else 0)       # the first field of this frame is a reference of the second
                is_p = avail > 0 and r.p(0.85)
                nact = max(1, min(avail, r.i(1, 3)))
                cuts = [0] + sorted(set(r.i(1, nmb - 1) for _ in range(self.nslices - 1))) + [nmb]
                for s_ in range(len(cuts) - 1):
                    if cuts[s_] < cuts[s_ + 1]:
                        au += self.slice(i, i, idr, is_p, cuts[s_], cuts[s_ + 1], s_, nact, field=fld)
            units.append(au)
            nfr += 1
        self.mb_h = frame_h
        return units

    def build(self):
        if self.paff:
            return self.build_paff()
        if self.bmode:
            return self.build_b()
        units = []
        nmb = self.mb_w * self.mb_h
        frame_num, held = 0, 0
        for i in range(self.npics):
            idr = i == 0
            is_p = i > 0 and not (i == 4 and self.npics > 5)          # one more I picture (non-IDR) in the middle
            au = b""
            if idr:
                au += self.param_sets()
            self.begin_picture()
            nact = min(i, max(1, self.nrefs)) if not self.mmco else min(held, self.nrefs)
            marking = ("long" if i == 2 else ("reset" if i == 6 else None)) if self.mmco else None
            cuts = [0] + sorted(set(self.r.i(1, nmb - 1) for _ in range(self.nslices - 1))) + [nmb]
            for s_ in range(len(cuts) - 1):
                if cuts[s_] < cuts[s_ + 1]:
                    au += self.slice(i, frame_num, idr, is_p and not (self.mixed and self.r.p(0.4)), cuts[s_], cuts[s_ + 1], s_, nact, marking=marking)
            units.append(au)
            frame_num += 1
            held = 1 if marking == "reset" else min(held + 1, max(1, self.nrefs))
            if marking == "reset":
                frame_num = 1                                # the picture counts as frame_num 0 from here on
            if self.gaps and i in (2, 5):
                frame_num += 1 + (i == 5)                    # one / two frames "lost": the decoder fills the gap with copies of the previous frame
        return units

    def build_b(self):
        """I0 P4 b2 P8 b6 ... in decoding order (numbers: picture order counts): the B pictures are not references and lie
        between two of their references"""
        units = []
        nmb = self.mb_w * self.mb_h
        order = [("I", 0)]
        k = 1
        while len(order) < self.npics:
            order.append(("P", 4 * k))
            if len(order) < self.npics:
                order.append(("B", 4 * k - 2))
            k += 1
        if isinstance(self, MbaffStream):
            nmb //= 2                                        # an MBAFF frame's slices are runs of pairs
        prev_ref_frame_num, nref_pics = -1, 0
        for i, (kind, poc) in enumerate(order):
            idr = i == 0
            is_ref = kind != "B"
            frame_num = 0 if idr else prev_ref_frame_num + 1
            au = b""
            if idr:
                au += self.param_sets()
            self.begin_picture()
            nact = min(nref_pics, max(1, self.nrefs))
            cuts = [0] + sorted(set(self.r.i(1, max(1, nmb - 1)) for _ in range(self.nslices - 1))) + [nmb]
            for s_ in range(len(cuts) - 1):
                if cuts[s_] < cuts[s_ + 1]:
                    au += self.slice(i, frame_num, idr, kind == "P", cuts[s_], cuts[s_ + 1], s_, nact, is_b=kind == "B", poc=poc, ref_idc=2 if is_ref else 0)
            units.append(au)
            if is_ref:
                prev_ref_frame_num = frame_num
                nref_pics += 1
        return units


class MbaffStream(Stream):
    """MBAFF frames (mb_adaptive_frame_field_flag): macroblock PAIRS, each coded as two frame or two field macroblocks.  The grids
    keep one entry per macroblock (row 2 * pair row + position in the pair) in the macroblock's own block order; what changes is
    WHERE the neighbours A (left) and B (above) of a block are (6.4.12.2, Table 6-4).  No skipped macroblocks (the inference rules
    for mb_field_decoding_flag stay out of the picture), no modes that need the above-left neighbour macroblock."""

    def sps(self):
        w = Bits()
        profile = 244 if self.cidc == 3 else (122 if self.cidc == 2 else (110 if self.depth > 8 else 100))
        w.u(8, profile); w.u(8, 0); w.u(8, 40)
        w.ue(0)
        w.ue(self.cidc)
        if self.cidc == 3:
            w.u(1, 0)
        w.ue(self.depth - 8); w.ue(self.depth - 8); w.u(1, 0); w.u(1, 0)
        w.ue(0)
        if self.bmode:
            w.ue(0); w.ue(2)                                 # pic_order_cnt_type 0, 6 bits of pic_order_cnt_lsb
        else:
            w.ue(2)
        w.ue(max(1, self.nrefs)); w.u(1, 0)
        w.ue(self.mb_w - 1); w.ue(self.mb_h // 2 - 1)
        w.u(1, 0); w.u(1, 1); w.u(1, 1)                      # frame_mbs_only 0, mb_adaptive_frame_field 1, direct_8x8_inference
        w.u(1, 0)
        if self.bmode:                                       # VUI: bitstream restrictions alone (one picture of reordering), as in Stream.sps
            w.u(1, 1)
            for _ in range(8):
                w.u(1, 0)
            w.u(1, 1)
            w.u(1, 1); w.ue(0); w.ue(0); w.ue(16); w.ue(16); w.ue(1); w.ue(max(2, self.nrefs))
        else:
            w.u(1, 0)
        w.trailing()
        return nal(3, 7, w.bytes())

    def begin_picture(self):
        super().begin_picture()
        self.fld = np.zeros((self.mb_h // 2, self.mb_w), np.int64)       # per pair: coded as field macroblocks

    def pair_ok(self, px, py, sid):
        return 0 <= px < self.mb_w and 0 <= py < self.mb_h // 2 and self.slice_of[2 * py, px] == sid

    def nbrs(self, x, y, bw, bh, mbx, mby, sid):
        bx, by = x % bw, y % bh
        px, py, pos = mbx, mby >> 1, mby & 1
        cur_field = int(self.fld[py, px])
        if bx > 0:
            a = (y, x - 1)
        elif not self.pair_ok(px - 1, py, sid):
            a = None
        else:
            lf = int(self.fld[py, px - 1])
            max_h, yn = 4 * bh, 4 * by
            if not cur_field:
                if not lf:
                    nb_pos, ym = pos, yn
                else:
                    nb_pos, ym = (0 if yn % 2 == 0 else 1), ((yn + max_h) >> 1 if pos else yn >> 1)
            else:
                if not lf:
                    if yn < max_h // 2:
                        nb_pos, ym = 0, (yn << 1) + pos
                    else:
                        nb_pos, ym = 1, (yn << 1) + pos - max_h
                else:
                    nb_pos, ym = pos, yn
            a = (bh * (2 * py + nb_pos) + (ym >> 2), bw * (px - 1) + bw - 1)
        if by > 0:
            b = (y - 1, x)
        elif not cur_field and pos == 1:
            b = (bh * (2 * py) + bh - 1, x)                                  # the top macroblock of the same pair
        elif not self.pair_ok(px, py - 1, sid):
            b = None
        else:
            above_field = int(self.fld[py - 1, px])
            nb_pos = 0 if (cur_field and pos == 0 and above_field) else 1
            b = (bh * (2 * (py - 1) + nb_pos) + bh - 1, x)
        return a, b

    def usable(self, mbx, mby):
        """constrained_intra_pred: an inter macroblock's samples are not used"""
        return not self.cip or self.kind[mby][mbx] in ("i4", "i8", "i16", "pcm")

    def left_rows(self, mbx, mby, sid, left):
        """what the reference's fill_decode_caches derives (h264_mvpred.h:  left_samples_available / intra4x4_pred_mode_cache): pairs of the
        same kind -> the macroblock at the same position; a field macroblock next to a frame pair -> its upper eight rows look at the
        pair's top macroblock, the lower eight at the bottom one; a frame macroblock next to a field pair -> samples need BOTH field
        macroblocks, the predicted mode looks at the top one"""
        px, py, pos = mbx, mby >> 1, mby & 1
        if not self.pair_ok(px - 1, py, sid):
            return [False] * 4, [False] * 4
        cur, lf = int(self.fld[py, px]), int(self.fld[py, px - 1])
        u = [self.usable(px - 1, 2 * py), self.usable(px - 1, 2 * py + 1)]
        if cur == lf:
            return [u[pos]] * 4, [u[pos]] * 4
        if cur:
            rows = [u[0], u[0], u[1], u[1]]
            return rows, list(rows)
        return [u[0] and u[1]] * 4, [u[0]] * 4

    def intra_avail(self, mbx, mby, sid):
        px, py, pos = mbx, mby >> 1, mby & 1
        left = all(self.left_rows(mbx, mby, sid, None)[0])
        if pos == 1 and not self.fld[py, px]:
            top = self.usable(px, 2 * py)
        elif not self.pair_ok(px, py - 1, sid):
            top = False
        else:
            nb_pos = 0 if (self.fld[py, px] and pos == 0 and self.fld[py - 1, px]) else 1
            top = self.usable(px, 2 * (py - 1) + nb_pos)
        return left, top, False

    def ref_range(self, nact):
        return 2 * nact if self.cur_field else nact

    def slice(self, idx, frame_num, idr, is_p, first_pair, last_pair, sid, nact, is_b=False, poc=None, ref_idc=3, **kw):
        r = self.r
        w = Bits()
        w.ue(first_pair)                                     # first_mb_in_slice counts pairs in an MBAFF frame
        w.ue(6 if is_b else (5 if is_p else 7))
        w.ue(0)
        w.u(4, frame_num & 15)
        w.u(1, 0)                                            # field_pic_flag
        if idr:
            w.ue(idx & 3)
        if poc is not None:
            w.u(6, poc & 63)
        if is_b:
            w.u(1, r.i(0, 1))                                # direct_spatial_mv_pred_flag
            w.u(1, 1)
            w.ue(nact - 1); w.ue(nact - 1)
            w.u(1, 0); w.u(1, 0)                             # no reference list modification, either list
            if self.bmode == 2:
                self.weight_table(w, nact, 2)
        if is_p:
            w.u(1, 1)
            w.ue(nact - 1)
            w.u(1, 0)
            if self.weighted:
                self.weight_table(w, nact, 1)
        if idr:
            w.u(1, 0); w.u(1, 0)
        elif ref_idc:
            w.u(1, 0)
        self.qp = 26 + (0 if idx == 0 else r.i(-6, 6))
        w.se(self.qp - 26)
        idc = self.deblock_idc if self.deblock_idc >= 0 else r.i(0, 2)
        w.ue(idc)
        if idc != 1:
            w.se(r.i(-2, 2)); w.se(r.i(-2, 2))
        for p_ in range(first_pair, last_pair):
            px, py = p_ % self.mb_w, p_ // self.mb_w
            self.fld[py, px] = self.cur_field = int(r.p(0.5))
            for pos in (0, 1):
                mby = 2 * py + pos
                self.slice_of[mby, px] = sid
                if is_p or is_b:
                    w.ue(0)                                  # mb_skip_run
                if pos == 0:
                    w.u(1, self.cur_field)                   # mb_field_decoding_flag
                if not (is_p or is_b) or r.p(0.3 if is_p else 0.15):
                    self.intra_mb(w, px, mby, sid, 5 if is_p else (23 if is_b else 0))
                elif is_b:
                    self.b_mb(w, px, mby, sid, nact)
                else:
                    self.inter_mb(w, px, mby, sid, nact)
        w.trailing()
        return nal(ref_idc, 5 if idr else 1, w.bytes())

    def build(self):
        if self.bmode:
            return self.build_b()
        units = []
        npairs = self.mb_w * self.mb_h // 2
        for i in range(self.npics):
            idr = i == 0
            is_p = i > 0 and i != 4
            au = self.param_sets() if idr else b""
            self.begin_picture()
            nact = min(i, max(1, self.nrefs))
            cuts = [0] + sorted(set(self.r.i(1, npairs - 1) for _ in range(self.nslices - 1))) + [npairs]
            for s_ in range(len(cuts) - 1):
                if cuts[s_] < cuts[s_ + 1]:
                    au += self.slice(i, i, idr, is_p, cuts[s_], cuts[s_ + 1], s_, nact)
            units.append(au)
        return units


STREAMS = {
    # 8-bit 4:2:0 — also decoded through the Tier-2 bridge and sessions (tests/test_synth_streams.py)
    "420_8_slices": dict(mb_w=6, mb_h=4, chroma_idc=1, depth=8, seed=11, nslices=3, deblock_idc=2, nrefs=3, npics=7),
    "420_8_oneslice": dict(mb_w=5, mb_h=4, chroma_idc=1, depth=8, seed=12, nslices=1, deblock_idc=0, nrefs=2, npics=6),
    "420_8_qcif": dict(mb_w=11, mb_h=9, chroma_idc=1, depth=8, seed=21, nslices=4, deblock_idc=0, nrefs=4, npics=10, far=40),
    "420_8_b_implicit": dict(mb_w=7, mb_h=5, chroma_idc=1, depth=8, seed=31, nslices=2, deblock_idc=0, nrefs=3, npics=9, bmode=1, far=20),
    "420_8_b_explicit": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=8, seed=32, nslices=1, deblock_idc=0, nrefs=2, npics=7, bmode=2),
    "420_8_b_average": dict(mb_w=6, mb_h=4, chroma_idc=1, depth=8, seed=33, nslices=3, deblock_idc=2, nrefs=3, npics=7, bmode=3, weighted=False),
    "420_8_t8x8": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=8, seed=41, nslices=3, deblock_idc=0, nrefs=3, npics=9, bmode=1, far=20, t8x8=True),
    "420_8_nofilter": dict(mb_w=7, mb_h=5, chroma_idc=1, depth=8, seed=22, nslices=2, deblock_idc=1, nrefs=2, npics=6, weighted=False),
    # the profiles no offline clip has: Tier 1 inside the reference decoder
    "422_8": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=8, seed=13, nslices=2, deblock_idc=0, nrefs=2, npics=6),
    "422_8_b": dict(mb_w=8, mb_h=6, chroma_idc=2, depth=8, seed=23, nslices=3, deblock_idc=2, nrefs=3, npics=8, far=30),
    "420_10": dict(mb_w=5, mb_h=4, chroma_idc=1, depth=10, seed=14, nslices=1, deblock_idc=0, nrefs=2, npics=6),
    "420_10_b": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=10, seed=24, nslices=3, deblock_idc=2, nrefs=3, npics=8, far=30),
    "422_10": dict(mb_w=4, mb_h=4, chroma_idc=2, depth=10, seed=15, nslices=2, deblock_idc=2, nrefs=2, npics=6),
    "420_9": dict(mb_w=4, mb_h=3, chroma_idc=1, depth=9, seed=16, nslices=1, deblock_idc=0, nrefs=1, npics=5),
    "422_8_t8x8": dict(mb_w=6, mb_h=5, chroma_idc=2, depth=8, seed=42, nslices=2, deblock_idc=0, nrefs=2, npics=7, bmode=3, weighted=False, t8x8=True),
    "420_10_t8x8": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=10, seed=43, nslices=2, deblock_idc=2, nrefs=2, npics=7, bmode=1, t8x8=True),
    "422_10_t8x8": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=44, nslices=1, deblock_idc=0, nrefs=2, npics=6, t8x8=True),
    "444_8": dict(mb_w=6, mb_h=4, chroma_idc=3, depth=8, seed=51, nslices=3, deblock_idc=2, nrefs=3, npics=7),
    "444_8_b_t8x8": dict(mb_w=7, mb_h=5, chroma_idc=3, depth=8, seed=52, nslices=2, deblock_idc=0, nrefs=2, npics=9, bmode=1, t8x8=True, far=20),
    "444_10": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=10, seed=53, nslices=2, deblock_idc=0, nrefs=2, npics=6, bmode=2),
    "420_8_cip_mixed": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=8, seed=81, nslices=4, deblock_idc=0, nrefs=3, npics=8, cip=True, mixed=True, t8x8=True),
    "420_8_cip_b": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=8, seed=82, nslices=2, deblock_idc=2, nrefs=2, npics=7, cip=True, bmode=1),
    "444_8_cip": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=83, nslices=2, deblock_idc=0, nrefs=2, npics=6, cip=True, mixed=True),
    "422_10_cip": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=84, nslices=2, deblock_idc=0, nrefs=2, npics=6, cip=True, mixed=True),
    # a second sequence with another picture size, bit depth or chroma format follows the first (new SPS + IDR): decoders re-initialise
    "420_8_resize": [dict(mb_w=6, mb_h=4, chroma_idc=1, depth=8, seed=91, nslices=2, deblock_idc=0, nrefs=2, npics=4),
                     dict(mb_w=4, mb_h=5, chroma_idc=1, depth=8, seed=92, nslices=1, deblock_idc=0, nrefs=2, npics=4, bmode=1),
                     dict(mb_w=7, mb_h=3, chroma_idc=1, depth=8, seed=93, nslices=2, deblock_idc=2, nrefs=2, npics=3)],
    "mixed_formats": [dict(mb_w=5, mb_h=4, chroma_idc=1, depth=8, seed=94, nslices=1, deblock_idc=0, nrefs=2, npics=3),
                      dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=95, nslices=1, deblock_idc=0, nrefs=2, npics=3),
                      dict(mb_w=4, mb_h=4, chroma_idc=3, depth=8, seed=96, nslices=2, deblock_idc=0, nrefs=2, npics=3),
                      dict(mb_w=5, mb_h=3, chroma_idc=1, depth=8, seed=97, nslices=1, deblock_idc=0, nrefs=2, npics=3)],
    # the reference's own limit is 32 slices per picture (MAX_SLICES, h264dec.h: beyond it the decoder warns and its per-slice
    # reference tables alias); the bridge holds 64
    "420_8_slices30": dict(mb_w=10, mb_h=8, chroma_idc=1, depth=8, seed=98, nslices=30, deblock_idc=2, nrefs=2, npics=5, bmode=1),
    # reference lists re-ordered per slice, a picture at two indices with different weights
    "420_8_reorder": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=8, seed=111, nslices=4, deblock_idc=0, nrefs=4, npics=9, reorder=True),
    "420_8_reorder_b": dict(mb_w=7, mb_h=5, chroma_idc=1, depth=8, seed=112, nslices=3, deblock_idc=0, nrefs=3, npics=9, bmode=2, reorder=True),
    "444_8_reorder_b": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=113, nslices=2, deblock_idc=0, nrefs=3, npics=7, bmode=1, reorder=True),
    # several picture parameter sets (chroma QP offsets and scaling lists differ; a picture picks one)
    "420_8_pps": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=8, seed=121, nslices=5, deblock_idc=0, nrefs=3, npics=8, npps=4, bmode=1),
    "420_8_scaling": dict(mb_w=7, mb_h=5, chroma_idc=1, depth=8, seed=122, nslices=3, deblock_idc=0, nrefs=2, npics=7, npps=3, scaling=True, t8x8=True),
    "444_8_scaling": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=123, nslices=2, deblock_idc=0, nrefs=2, npics=6, npps=2, scaling=True, t8x8=True, bmode=1),
    "422_10_scaling": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=124, nslices=2, deblock_idc=0, nrefs=2, npics=6, npps=2, scaling=True, t8x8=True),
    # frame_num gaps: the decoder inserts frames the bitstream never carried (copies of the previous one) and predicts from them
    "420_8_gaps": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=8, seed=131, nslices=2, deblock_idc=0, nrefs=3, npics=9, gaps=True),
    # every slice with its own disable_deblocking_filter_idc; a picture of one macroblock row.  (Pictures ONE MACROBLOCK WIDE are
    # not used: the reference's own decoder is not self-consistent there — with all vectors zero a P macroblock with 4-wide
    # partitions does not come out as a copy of its reference, while its DSP functions called one by one are right; this
    # project's hooks, bridge and oracle agree with each other and with the standard on such streams.)
    "420_8_idc_per_slice": dict(mb_w=8, mb_h=6, chroma_idc=1, depth=8, seed=141, nslices=6, deblock_idc=-1, nrefs=2, npics=8, bmode=1),
    "420_8_9x1": dict(mb_w=9, mb_h=1, chroma_idc=1, depth=8, seed=143, nslices=2, deblock_idc=0, nrefs=2, npics=6, bmode=1),
    "444_8_2x7": dict(mb_w=2, mb_h=7, chroma_idc=3, depth=8, seed=144, nslices=2, deblock_idc=-1, nrefs=2, npics=6),
    # adaptive reference marking: a long-term reference frame, then operation 5 (everything unused, frame_num and POC start over)
    "420_8_mmco": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=8, seed=151, nslices=2, deblock_idc=0, nrefs=4, npics=10, mmco=True),
    # interlaced-capable sequences (frame_mbs_only_flag 0): frame pictures and field pairs mixed
    "420_8_paff": dict(mb_w=6, mb_h=6, chroma_idc=1, depth=8, seed=101, nslices=2, deblock_idc=0, nrefs=2, npics=8, paff=True),
    "422_10_paff": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=102, nslices=1, deblock_idc=0, nrefs=2, npics=6, paff=True, t8x8=True),
    "444_8_paff": dict(mb_w=4, mb_h=4, chroma_idc=3, depth=8, seed=103, nslices=2, deblock_idc=2, nrefs=2, npics=6, paff=True, cip=True),
    "420_8_paff_b": dict(mb_w=8, mb_h=8, chroma_idc=1, depth=8, seed=104, nslices=4, deblock_idc=-1, nrefs=4, npics=12, paff=True, cip=True, far=24),
    "420_8_paff_t8x8": dict(mb_w=7, mb_h=6, chroma_idc=1, depth=8, seed=105, nslices=3, deblock_idc=2, nrefs=3, npics=10, paff=True, t8x8=True, npps=3, scaling=True),
    "paff_and_frames": [dict(mb_w=6, mb_h=6, chroma_idc=1, depth=8, seed=106, nslices=2, deblock_idc=0, nrefs=2, npics=5, paff=True),
                        dict(mb_w=6, mb_h=6, chroma_idc=1, depth=8, seed=107, nslices=1, deblock_idc=0, nrefs=2, npics=4, bmode=1),
                        dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=108, nslices=2, deblock_idc=0, nrefs=2, npics=5, paff=True),
                        dict(mb_w=5, mb_h=4, chroma_idc=2, depth=8, seed=109, nslices=1, deblock_idc=0, nrefs=2, npics=3, paff=True)],
    # MBAFF: macroblock pairs coded as frame or field macroblocks (Tier 1: the *_mbaff loop filters; Tier 2 steps aside)
    "420_8_mbaff": dict(mb_w=7, mb_h=6, chroma_idc=1, depth=8, seed=161, nslices=2, deblock_idc=0, nrefs=2, npics=7, mbaff=True),
    "422_10_mbaff": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=162, nslices=1, deblock_idc=0, nrefs=2, npics=6, mbaff=True),
    "444_8_mbaff": dict(mb_w=4, mb_h=4, chroma_idc=3, depth=8, seed=163, nslices=2, deblock_idc=-1, nrefs=2, npics=6, mbaff=True),
    # ... with B pictures: implicit weights (field macroblocks weigh by the distances between FIELDS), explicit tables, the plain average
    "420_8_mbaff_b_implicit": dict(mb_w=6, mb_h=4, chroma_idc=1, depth=8, seed=164, nslices=2, deblock_idc=0, nrefs=2, npics=7, mbaff=True, bmode=1),
    "420_10_mbaff_b_explicit": dict(mb_w=5, mb_h=4, chroma_idc=1, depth=10, seed=165, nslices=1, deblock_idc=-1, nrefs=2, npics=7, mbaff=True, bmode=2, weighted=True),
    "422_8_mbaff_b": dict(mb_w=4, mb_h=4, chroma_idc=2, depth=8, seed=166, nslices=2, deblock_idc=0, nrefs=3, npics=7, mbaff=True, bmode=3),
    # ... with constrained intra prediction: a field macroblock beside a frame pair of which one macroblock is inter sees HALF a left edge
    # (the reference's extra chroma DC predictors, h264pred_template.c:716-766), a frame macroblock beside a field pair needs both
    "420_8_mbaff_cip": dict(mb_w=7, mb_h=6, chroma_idc=1, depth=8, seed=167, nslices=2, deblock_idc=0, nrefs=2, npics=7, mbaff=True, cip=True, t8x8=True),
    "422_10_mbaff_cip_b": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=168, nslices=2, deblock_idc=-1, nrefs=2, npics=7, mbaff=True, cip=True, bmode=1),
    "420_8_cropped": dict(mb_w=6, mb_h=5, chroma_idc=1, depth=8, seed=71, nslices=2, deblock_idc=0, nrefs=2, npics=7, bmode=1, crop=(3, 4)),
    "444_8_cropped": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=72, nslices=1, deblock_idc=0, nrefs=2, npics=5, crop=(5, 7)),
    "422_10_cropped": dict(mb_w=5, mb_h=4, chroma_idc=2, depth=10, seed=73, nslices=1, deblock_idc=0, nrefs=2, npics=5, crop=(2, 9)),
    "420_8_lossless": dict(mb_w=5, mb_h=4, chroma_idc=1, depth=8, seed=61, nslices=2, deblock_idc=0, nrefs=2, npics=6, weighted=False, lossless=True),
    "444_8_lossless": dict(mb_w=5, mb_h=4, chroma_idc=3, depth=8, seed=62, nslices=1, deblock_idc=0, nrefs=2, npics=6, weighted=False, lossless=True, t8x8=True),
    "422_10_lossless": dict(mb_w=4, mb_h=4, chroma_idc=2, depth=10, seed=63, nslices=1, deblock_idc=0, nrefs=2, npics=5, weighted=False, lossless=True),
    # the two places where the reference's decoder is not consistent with itself: pictures two macroblocks wide with two-reference weighted
    # prediction (h264_mb.c:407-409: the Cr intermediate overwrites the Cb one), field pictures with disable_deblocking_filter_idc 2 and intra macroblocks
    # (h264_mb.c:525-527: the other field's row of slice_table decides the above-left border).  The bridge leaves such pictures to the C path.
    "420_8_2wide_b": dict(mb_w=2, mb_h=5, chroma_idc=1, depth=8, seed=171, nslices=1, deblock_idc=0, nrefs=2, npics=7, bmode=2),
    "420_8_paff_idc2_intra": dict(mb_w=6, mb_h=6, chroma_idc=1, depth=8, seed=172, nslices=2, deblock_idc=2, nrefs=2, npics=8, paff=True, cip=True, mixed=True),
    "422_8_bframes": dict(mb_w=6, mb_h=4, chroma_idc=2, depth=8, seed=34, nslices=2, deblock_idc=0, nrefs=2, npics=7, bmode=1),
    "420_10_bframes": dict(mb_w=6, mb_h=4, chroma_idc=1, depth=10, seed=35, nslices=2, deblock_idc=0, nrefs=2, npics=7, bmode=2),
}



def write_samples(path, units):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0) + struct.pack("<I", len(units)))
        for u in units:
            f.write(struct.pack("<I", len(u)) + u)


def decode_plain(samples, out):
    """the reference decoder, its tables untouched (oracle/_ref/h264_tier1_emu with MI355_TIER1_PLAIN); returns its stderr"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "h264_tier1_emu")
    r = subprocess.run([exe, samples, out], capture_output=True, text=True, env=dict(os.environ, MI355_TIER1_PLAIN="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def main():
    import hashlib
    import json
    import tempfile
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))
    T = load_tables()
    md5 = {}
    for name, kw in STREAMS.items():
        def make(k):
            k = dict(k)
            return (MbaffStream if k.pop("mbaff", False) else Stream)(T, name, **k).build()
        units = sum((make(k) for k in kw), []) if isinstance(kw, list) else make(kw)
        p = os.path.join(out, "h264_synth_%s.samples" % name)
        write_samples(p, units)
        with tempfile.TemporaryDirectory() as td:
            err = decode_plain(p, os.path.join(td, "o.yuv"))
            lines = [l for l in err.splitlines() if l.strip()]
            assert len(lines) == 1 and ("%d pictures" % len(units)) in lines[0], err      # nothing but the harness's summary: no decoder complaint
            raw = open(os.path.join(td, "o.yuv"), "rb").read()
        md5[name] = dict(pictures=len(units), bytes=len(raw), md5=hashlib.md5(raw).hexdigest(), summary=lines[0].split("hooked, ")[1])
        print(name, len(units), "pictures", sum(map(len, units)), "bytes ->", md5[name]["summary"], md5[name]["md5"])
    json.dump(md5, open(os.path.join(out, "h264_synth_ref_md5.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
