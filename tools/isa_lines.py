#!/usr/bin/env python3
"""Static instruction counts of a kernel per source line (hipcc -S -gline-tables-only): which statements the VALU / SALU /
LDS / VMEM instructions were generated for.  usage: isa_lines.py file.s kernel_substring [min_count]"""
import collections
import re
import sys


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kname = sys.argv[1], sys.argv[2]
    minc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    files, cur, inside = {}, None, False
    cnt = collections.defaultdict(collections.Counter)
    for l in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if re.match(r"^_Z\w*%s\w*:" % re.escape(kname), l):
            inside = True
            continue
        if inside and l.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = "%s:%d" % (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m:
            cnt[cur][kind(m.group(1))] += 1
    tot = collections.Counter()
    for loc, c in sorted(cnt.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"])):
        tot.update(c)
        if c["valu"] + c["salu"] + c["lds"] >= minc:
            print("%-24s valu %4d salu %4d lds %3d vmem %3d wait %3d" % (loc, c["valu"], c["salu"], c["lds"], c["vmem"], c["wait"]))
    print("TOTAL", dict(tot))


if __name__ == "__main__":
    main()
