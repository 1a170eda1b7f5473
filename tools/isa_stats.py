#!/usr/bin/env python3
"""Static instruction mix of a gfx950 kernel from hipcc -S output: per basic block, counts of VALU / SALU / LDS /
VMEM / waits, so that the hot loop's issue cost can be read without a GPU run.
usage: isa_stats.py file.s kernel_substring [min_block_size]"""
import re
import sys


def classify(op):
    if op.startswith(("v_cmp", "v_")):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kname = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(kname), l):
            start = i
            break
    assert start is not None, "kernel not found"
    blocks, cur, name = [], {}, "entry"
    total = {}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.startswith("\t.section"):
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), {}
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or l.startswith("\t."):
            continue
        c = classify(m.group(1))
        cur[c] = cur.get(c, 0) + 1
        total[c] = total.get(c, 0) + 1
    blocks.append((name, cur))
    keys = ["valu", "salu", "lds", "vmem", "smem", "wait", "branch", "nop"]
    print("%-14s" % "block" + "".join("%7s" % k for k in keys))
    for n, b in blocks:
        if sum(b.values()) >= minsz:
            print("%-14s" % n + "".join("%7d" % b.get(k, 0) for k in keys))
    print("%-14s" % "TOTAL" + "".join("%7d" % total.get(k, 0) for k in keys))


if __name__ == "__main__":
    main()
