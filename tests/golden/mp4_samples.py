#!/usr/bin/env python3
"""Minimal MP4 reader: pull the H.264 track's avcC blob and its samples (decode order)
out of an .mp4 so the reference decoder can be fed without libavformat.
Writes: u32 extradata_len, extradata, u32 n, n x {u32 len, bytes}."""
import struct
import sys


def boxes(buf, start, end):
    p = start
    while p + 8 <= end:
        size, typ = struct.unpack(">I4s", buf[p:p + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[p + 8:p + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - p
        yield typ, p + hdr, p + size
        p += size


def find(buf, start, end, path):
    for typ, s, e in boxes(buf, start, end):
        if typ == path[0]:
            if len(path) == 1:
                yield s, e
            else:
                yield from find(buf, s, e, path[1:])


def extract(path):
    buf = open(path, "rb").read()
    for ts, te in find(buf, 0, len(buf), [b"moov", b"trak"]):
        stbl = list(find(buf, ts, te, [b"mdia", b"minf", b"stbl"]))
        if not stbl:
            continue
        ss, se = stbl[0]
        stsd = list(find(buf, ss, se, [b"stsd"]))[0]
        entry = stsd[0] + 8                     # version/flags + entry count
        esize, etype = struct.unpack(">I4s", buf[entry:entry + 8])
        if etype != b"avc1":
            continue
        avcc = None
        for typ, s, e in boxes(buf, entry + 8 + 78, entry + esize):   # VisualSampleEntry is 78 bytes
            if typ == b"avcC":
                avcc = buf[s:e]
        s, e = list(find(buf, ss, se, [b"stsz"]))[0]
        uniform, count = struct.unpack(">II", buf[s + 4:s + 12])
        sizes = [uniform] * count if uniform else list(struct.unpack(">%dI" % count, buf[s + 12:s + 12 + 4 * count]))
        co = list(find(buf, ss, se, [b"stco"]))
        if co:
            s, e = co[0]
            n = struct.unpack(">I", buf[s + 4:s + 8])[0]
            offs = list(struct.unpack(">%dI" % n, buf[s + 8:s + 8 + 4 * n]))
        else:
            s, e = list(find(buf, ss, se, [b"co64"]))[0]
            n = struct.unpack(">I", buf[s + 4:s + 8])[0]
            offs = list(struct.unpack(">%dQ" % n, buf[s + 8:s + 8 + 8 * n]))
        s, e = list(find(buf, ss, se, [b"stsc"]))[0]
        n = struct.unpack(">I", buf[s + 4:s + 8])[0]
        stsc = [struct.unpack(">III", buf[s + 8 + 12 * i:s + 20 + 12 * i]) for i in range(n)]
        samples, si = [], 0
        for ci, off in enumerate(offs):
            per = [x for x in stsc if x[0] <= ci + 1][-1][1]
            p = off
            for _ in range(per):
                if si >= count:
                    break
                samples.append(buf[p:p + sizes[si]])
                p += sizes[si]
                si += 1
        return avcc, samples
    raise SystemExit("no avc1 track")


if __name__ == "__main__":
    avcc, samples = extract(sys.argv[1])
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else len(samples)
    samples = samples[:limit]
    with open(sys.argv[2], "wb") as f:
        f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
        for s in samples:
            f.write(struct.pack("<I", len(s)) + s)
    print("avcC %d bytes, %d samples" % (len(avcc), len(samples)))
