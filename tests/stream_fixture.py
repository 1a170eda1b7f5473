"""Real-bitstream fixture: Tier-2 records exported from the REFERENCE decoder's own run
(oracle/ref_h264_export.c) + the reference's decoded pictures.

parse_export(bin)  -> list of pictures (dicts) straight from the exporter's output
save_npz / load_npz -> the committed, compressed form (tests/golden/h264_stream_*.npz)
frameset_for(pics, i) -> a one-picture h264_frames.FrameSet whose reference slots hold the
                         reference decoder's own earlier output pictures
"""
import struct

import numpy as np

import h264_frames as HF


def parse_export(path):
    buf = open(path, "rb").read()
    p, pics = 0, []
    while p < len(buf):
        magic, mbw, mbh, nsl, nslots, l1, maxl, ptype = struct.unpack_from("<8I", buf, p)
        assert magic == 0x46523634
        p += 32
        slots = struct.unpack_from("<%dI" % HF.MAX_SLOTS, buf, p)
        p += 4 * HF.MAX_SLOTS
        nmb = mbw * mbh
        mb = np.frombuffer(buf, HF.MB_DT, nmb, p).copy(); p += nmb * 64
        mv0 = np.frombuffer(buf, np.int16, nmb * 32, p).reshape(nmb, 16, 2).copy(); p += nmb * 64
        mv1 = np.frombuffer(buf, np.int16, nmb * 32, p).reshape(nmb, 16, 2).copy(); p += nmb * 64
        coef = np.frombuffer(buf, np.int16, nmb * 384, p).reshape(nmb, 384).copy(); p += nmb * 768
        sl = np.frombuffer(buf, HF.SLICE_DT, nsl, p).copy(); p += nsl * HF.SLICE_DT.itemsize
        W, H = 16 * mbw, 16 * mbh
        y = np.frombuffer(buf, np.uint8, W * H, p).reshape(H, W).copy(); p += W * H
        cb = np.frombuffer(buf, np.uint8, W * H // 4, p).reshape(H // 2, W // 2).copy(); p += W * H // 4
        cr = np.frombuffer(buf, np.uint8, W * H // 4, p).reshape(H // 2, W // 2).copy(); p += W * H // 4
        pics.append(dict(mb_w=mbw, mb_h=mbh, slots=[s for s in slots[:nslots]], use_l1=bool(l1), pict_type=ptype,
                         mb=mb, mv0=mv0, mv1=mv1, coef=coef, slices=sl, y=y, cb=cb, cr=cr))
    return pics


def save_npz(path, pics):
    d = {"n": np.array(len(pics))}
    for i, pc in enumerate(pics):
        for k, v in pc.items():
            d["%d_%s" % (i, k)] = np.asarray(v)
    np.savez_compressed(path, **d)


def load_npz(path):
    z = np.load(path)
    pics = []
    for i in range(int(z["n"])):
        pc = {}
        for k in ("mb_w", "mb_h", "slots", "use_l1", "pict_type", "mb", "mv0", "mv1", "coef", "slices", "y", "cb", "cr"):
            v = z["%d_%s" % (i, k)]
            pc[k] = v if v.ndim else v.item()
        pc["slots"] = [int(s) for s in np.atleast_1d(pc["slots"])]
        pc["mb"] = pc["mb"].view(HF.MB_DT).reshape(-1) if pc["mb"].dtype != HF.MB_DT else pc["mb"]
        pc["slices"] = HF.as_slices(pc["slices"]).reshape(-1)
        pics.append(pc)
    return pics


def frameset_for(pics, i):
    pc = pics[i]
    nslots = max(1, len(pc["slots"]))
    fs = HF.FrameSet(1, pc["mb_w"], pc["mb_h"], nslots)
    fs.mb[0] = pc["mb"]
    fs.mv[0, 0] = pc["mv0"]
    fs.mv[1, 0] = pc["mv1"]
    fs.coef[0] = pc["coef"]
    fs.slices = pc["slices"].reshape(1, -1).copy()
    fs.use_l1 = bool(pc["use_l1"])
    for s in range(nslots):
        if s < len(pc["slots"]):
            r = pics[pc["slots"][s]]
            fs.refs[0][s] = (r["y"], r["cb"], r["cr"])
        else:
            fs.refs[0][s] = (np.zeros_like(pc["y"]), np.zeros_like(pc["cb"]), np.zeros_like(pc["cr"]))
    fs.max_intra_level = HF.intra_schedule(fs, 0)
    return fs


def frameset_all(pics, first=0, count=None):
    """All pictures as ONE batch of independent pictures (each picture's reference slots hold the
    reference decoder's own output, so nothing depends on another picture of the batch)."""
    count = count or len(pics) - first
    sel = pics[first:first + count]
    nslots = max(1, max(len(p["slots"]) for p in sel))
    fs = HF.FrameSet(len(sel), sel[0]["mb_w"], sel[0]["mb_h"], nslots)
    nsl = max(len(p["slices"]) for p in sel)
    fs.slices = np.zeros((len(sel), nsl), HF.SLICE_DT)
    blank = None
    for f, pc in enumerate(sel):
        fs.mb[f] = pc["mb"]
        fs.mv[0, f] = pc["mv0"]
        fs.mv[1, f] = pc["mv1"]
        fs.coef[f] = pc["coef"]
        fs.slices[f, :len(pc["slices"])] = pc["slices"]
        fs.use_l1 = fs.use_l1 or bool(pc["use_l1"])
        for s in range(nslots):
            if s < len(pc["slots"]):
                r = pics[pc["slots"][s]]
                fs.refs[f][s] = (r["y"], r["cb"], r["cr"])
            else:
                if blank is None:
                    blank = (np.zeros_like(pc["y"]), np.zeros_like(pc["cb"]), np.zeros_like(pc["cr"]))
                fs.refs[f][s] = blank
        fs.max_intra_level = max(fs.max_intra_level, HF.intra_schedule(fs, f))
    return fs
