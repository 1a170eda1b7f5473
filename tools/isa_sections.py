#!/usr/bin/env python3
"""Static instruction counts of a kernel's listing between the `; MARK name` comments an analysis build (-DFQ_MARKS) leaves in it, in layout order.
usage: tools/isa_sections.py build/isa/<file>.s <kernel substring>"""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
inside, sec = False, "head"
cnt = collections.OrderedDict()
for line in open(path):
    t = line.strip()
    if not inside:
        if re.match(r"[A-Za-z_][\w$.]*:", t) and kern in t.split(":")[0]:
            inside = True
        continue
    if t.startswith("s_endpgm"):
        break
    m = re.match(r"; MARK (\w+)", t)
    if m:
        sec = m.group(1)
        continue
    op = t.split()[0] if t and not t.startswith((";", ".")) and not re.match(r"[\w$.]+:", t) else None
    if not op:
        continue
    c = cnt.setdefault(sec, collections.Counter())
    if op.startswith("v_mfma"): c["mfma"] += 1
    elif op.startswith("v_"): c["valu"] += 1
    elif op.startswith(("s_load", "s_buffer")): c["smem"] += 1
    elif op.startswith(("s_waitcnt", "s_nop")): c["wait"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
    else: c["other"] += 1
keys = ["valu", "mfma", "salu", "branch", "smem", "wait", "lds", "vmem"]
print("%-14s" % "section" + "".join("%8s" % k for k in keys))
for sec, c in cnt.items():
    print("%-14s" % sec + "".join("%8d" % c[k] for k in keys))
