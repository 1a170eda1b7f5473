"""include/mi355_h264_frame.h against the Python mirrors the tests fill (tests/h264_frames.py): sizes and offsets of the macroblock record, the
slice record (with the field-macroblock implicit weights of MBAFF slices) and the picture descriptor, read from a C probe compiled here."""
import ctypes as C
import os
import subprocess
import tempfile

import h264_frames as HF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MB = {"mb_type": "mb_type", "nnz_mask": "nnz_mask", "cbp": "cbp", "qp": "qp", "flags": "flags", "alpha": "slice_alpha_c0_offset", "beta": "slice_beta_offset",
      "i16mode": "intra16x16_pred_mode", "chroma_mode": "chroma_pred_mode", "topleft": "topleft_samples_available", "topright": "topright_samples_available",
      "sub": "sub_mb_type", "ref_idx": "ref_idx", "dc_qmul": "dc_qmul", "slice_id": "slice_id", "intra_level": "intra_level", "qpc": "qpc", "i4mode": "u"}
SL = {"use_weight": "use_weight", "use_weight_chroma": "use_weight_chroma", "luma_denom": "luma_log2_weight_denom", "chroma_denom": "chroma_log2_weight_denom",
      "list_count": "list_count", "ref_slot": "ref_slot", "luma_weight": "luma_weight", "chroma_weight": "chroma_weight", "implicit_weight": "implicit_weight",
      "chroma_qp_table": "chroma_qp_table", "implicit_weight_field": "implicit_weight_field"}
FR = {"mb_width": "mb_width", "mb_height": "mb_height", "dst": "dst", "dst_stride": "dst_stride", "recon": "recon", "recon_stride": "recon_stride", "ref": "ref",
      "mb": "mb", "mv": "mv", "coef": "coef", "slices": "slices", "nslices": "nslices", "max_intra_level": "max_intra_level", "intra_list": "intra_list",
      "intra_level_start": "intra_level_start", "max_level_width": "max_level_width", "reserved": "field_picture", "surface_layout": "surface_layout", "flags": "flags"}


def _probe():
    lines = ['    printf("mb=%zu slice=%zu frame=%zu\\n", sizeof(mi355_h264_mb), sizeof(mi355_h264_slice), sizeof(mi355_h264_frame));']
    for tag, st, names in (("mb", "mi355_h264_mb", MB), ("slice", "mi355_h264_slice", SL), ("frame", "mi355_h264_frame", FR)):
        for py, c in names.items():
            lines.append('    printf("%s.%s=%%zu\\n", offsetof(%s, %s));' % (tag, py, st, c))
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"mi355_h264_frame.h\"\nint main(void) {\n%s\n    return 0;\n}\n" % "\n".join(lines)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")], check=True)
        out = subprocess.run([os.path.join(d, "p")], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes = dict(kv.split("=") for kv in out[0].split())
    offs = dict(ln.split("=") for ln in out[1:] if ln)
    return {k: int(v) for k, v in sizes.items()}, {k: int(v) for k, v in offs.items()}


def test_record_mirrors_match_the_header():
    sizes, offs = _probe()
    assert sizes == {"mb": HF.MB_DT.itemsize, "slice": HF.SLICE_DT.itemsize, "frame": C.sizeof(HF.Frame)}
    for py in MB:
        assert HF.MB_DT.fields[py][1] == offs["mb." + py], py
    for py in SL:
        assert HF.SLICE_DT.fields[py][1] == offs["slice." + py], py
    for py in FR:
        assert getattr(HF.Frame, py).offset == offs["frame." + py], py
    # every field of the mirrors is named above (a field added on one side only fails here)
    assert set(HF.MB_DT.names) == set(MB) and set(HF.SLICE_DT.names) - {"rsv"} == set(SL) and {f[0] for f in HF.Frame._fields_} == set(FR)
