#!/usr/bin/env python3
"""Static counts of a listing's kernels that say how a kernel gets at its records: vector loads, scalar loads, and the waits for ALL outstanding
vector loads (a `s_waitcnt vmcnt(0)` right behind a lone load is a dependent round trip).  tools/isa_loads.py build/isa/<file>.s [kernel substring]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2] if len(sys.argv) > 2 else ''
name, rows = None, []
def flush():
    if name is None or sub not in name: return
    ins = [l for l in rows if l and not l.startswith(('.', ';')) and not l.endswith(':')]
    c = lambda p: sum(1 for l in ins if l.startswith(p))
    w = lambda p: sum(1 for l in ins if l.startswith('s_waitcnt') and p in l)
    print(f"{name[:70]:70s} insts {len(ins):6d}  global_load {c('global_load'):4d}  s_load {c('s_load'):4d}  flat_load {c('flat_load'):4d}  calls {c('s_swappc'):3d}  ds {c('ds_'):5d}  vmcnt(0) {w('vmcnt(0)'):4d}  lgkmcnt(0) {w('lgkmcnt(0)'):4d}  scratch {c('scratch_'):3d}  branches {c('s_cbranch'):4d}")
for l in txt:
    m = re.match(r'^(_Z\w+):\s*;', l)
    if m:
        flush(); name, rows = m.group(1), []
    elif name is not None:
        rows.append(l.strip())
        if l.strip().startswith('.end_amdhsa_kernel') or l.strip() == '.cfi_endproc': pass
flush()
