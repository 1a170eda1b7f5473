#!/usr/bin/env python3
"""Where do the slow integer instructions of a kernel come from?  hipcc -S -gline-tables-only output: every quarter-rate
or 64-bit VALU instruction (v_mul_lo/hi_u32, v_mad_u64_u32, v_mad_i64_i32) with the source line it was generated for.
usage: isa_hot.py file.s kernel_substring"""
import collections
import re
import sys

SLOW = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_i32")


def main():
    path, kname = sys.argv[1], sys.argv[2]
    files, cur, inside = {}, None, False
    hits = collections.Counter()
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
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m and m.group(1).startswith(SLOW):
            hits[(cur, m.group(1).replace("_e32", "").replace("_e64", ""))] += 1
    for (loc, op), n in sorted(hits.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
        print("%-26s %-16s x%d" % ("%s:%s" % loc if loc else "?", op, n))
    print("total", sum(hits.values()))


if __name__ == "__main__":
    main()
