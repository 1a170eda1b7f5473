"""include/mi355_abi.h must be layout-identical to the reference's structs."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import abi_ctypes as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_layout_ref.json")))

PROBE = r'''
#include <stdio.h>
#include "mi355_abi.h"
#define F(S, f) printf(#S "." #f "=%%zu\n", offsetof(S, f))
#define Z(S)    printf(#S "=%%zu\n", sizeof(S))
int main(void) {
%s
    printf("AV_CODEC_ID_H264=%%d\n", MI355_AV_CODEC_ID_H264);
    return 0;
}
'''


def test_header_layout_matches_reference_probe():
    lines = []
    for key in GOLD:
        if key == "AV_CODEC_ID_H264":
            continue
        if "." in key:
            s, f = key.split(".")
            lines.append("    F(%s, %s);" % (s, f))
        else:
            lines.append("    Z(%s);" % key)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.c")
        open(src, "w").write(PROBE % "\n".join(lines))
        exe = os.path.join(d, "probe")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = {k: int(v) for k, v in (ln.split("=") for ln in out.split("\n") if ln)}
    assert got == GOLD


def test_ctypes_mirror_matches_reference_probe():
    for key, val in GOLD.items():
        if key == "AV_CODEC_ID_H264":
            assert A.AV_CODEC_ID_H264 == val
        elif "." in key:
            s, f = key.split(".")
            assert getattr(getattr(A, s), f).offset == val, key
        else:
            assert C.sizeof(getattr(A, key)) == val, key


def test_golden_layout_is_current(ref):
    buf = C.create_string_buffer(8192)
    ref.lib.ref_layout.restype = C.c_int
    n = ref.lib.ref_layout(buf, 8192)
    now = {k: int(v) for k, v in (ln.split("=") for ln in buf.raw[:n].decode().split("\n") if ln)}
    assert now == GOLD
