#!/usr/bin/env python3
"""Generate the SAO-decision golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_SAODEC_DUMP set, so the --wrap interposers of
oracle/ref_harness_saodec_dump.c record, per sampled LCU, the statistics SaoGenerationDecision(16bit) gathered, the lambdas,
rate tables and mode switches it read, the neighbours' parameters and everything it decided.
Whole pictures are kept (every LCU the encode pass ran the decision for), so the fixtures serve both the per-LCU oracle test
and the picture-level device test.  Stored as tests/golden/saodec_<name>.npz (a structured array of records + width, height).  Needs /root/reference (this container only).
Usage: python tests/golden/make_saodec_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

PARAMS = np.dtype([("merge_left", "u1"), ("merge_up", "u1"), ("pad", "u1", 2), ("type", "<u4", 2), ("offset", "<i4", (3, 4)),
                   ("band", "<u4", 3)])
REC = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("is16", "<u4"), ("mm_sao", "<u4"), ("temporal_layer", "<u4"),
                ("has_left", "<u4"), ("has_up", "<u4"), ("pad", "<u4"), ("picture_number", "<u8"), ("origin_x", "<u4"), ("origin_y", "<u4"),
                ("lambda", "<u8"), ("chroma_lambda", "<u8"),
                ("type_bits", "<u4", 6), ("merge_bits", "<u4", 2), ("offset_bits", "<u4", 8),
                ("left", PARAMS), ("up", PARAMS), ("out", PARAMS), ("luma_cost", "<i8"), ("chroma_cost", "<i8"),
                ("bo_diff", "<i4", (3, 32)), ("bo_count", "<u2", (3, 32)), ("eo_diff", "<i4", (3, 4, 5)), ("eo_count", "<u2", (3, 4, 5))],
               align=True)

# name -> (clip kind, width, height, frames, seed, bit depth, encoder args, whole pictures kept)
CASES = {
    "p_416x240_m9": ("motion", 416, 240, 5, 7, 8, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0"], 4),
    "b_416x240_m5_q30": ("motion", 416, 240, 9, 7, 8, ["-encMode", "5", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "30"], 5),
    "noise_320x200_m1_q45": ("noise", 320, 200, 5, 11, 8, ["-encMode", "1", "-pred-struct", "1", "-hierarchical-levels", "1", "-q", "45"], 4),
    "p10_416x240_m9_q20": ("motion", 416, 240, 4, 7, 10, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0", "-bit-depth", "10", "-q", "20"], 3),
    # saoMode 0 (the reduced luma-only decision) exists only at encMode 11, which needs a 4K input
    "b_3840x2160_m11": ("motion", 3840, 2160, 9, 7, 8, ["-encMode", "11", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "36"], 2),
    "b10_3840x2160_m11": ("motion", 3840, 2160, 5, 7, 10, ["-encMode", "11", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "18", "-bit-depth", "10"], 1),
    "tiles_640x384_m3": ("motion", 640, 384, 3, 7, 8, ["-encMode", "3", "-pred-struct", "0", "-hierarchical-levels", "0", "-tile_row_cnt", "2", "-tile_col_cnt", "2"], 2),
}


def run_case(name):
    kind, w, h, n, seed, depth, args, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "saodec.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-sao", "1", "-b", os.path.join(td, "out.265")]
        if "-q" not in args:
            cmd += ["-q", "32"]
        subprocess.run(cmd + args, env=dict(os.environ, SVT_REF_SAODEC_DUMP=dump, SVT_REF_SAODEC_STRIDE="1"), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, REC)
    assert len(recs) and (recs["magic"] == 0x44414f53).all() and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    # whole pictures, the ones with the most going on first: merges, band offsets, chroma on
    o = recs["out"]
    score = (o["merge_left"].astype(int) + o["merge_up"]) * 2 + (o["type"][:, 0] == 5) * 3 + (o["type"][:, 1] != 0) * 2 + (o["type"][:, 0] != 0) * 2
    pics = np.unique(recs["picture_number"])
    by_layer = {}
    for pn in sorted(pics, key=lambda pn: -int(score[recs["picture_number"] == pn].sum())):
        by_layer.setdefault(int(recs["temporal_layer"][recs["picture_number"] == pn][0]), []).append(pn)
    chosen = []
    while len(chosen) < min(keep, len(pics)):        # round-robin over the temporal layers
        for layer in sorted(by_layer):
            if by_layer[layer] and len(chosen) < keep:
                chosen.append(by_layer[layer].pop(0))
    out = recs[np.isin(recs["picture_number"], chosen)]
    out = out[np.lexsort((out["origin_x"], out["origin_y"], out["picture_number"]))]
    path = os.path.join(S.GOLDEN_DIR, "saodec_%s.npz" % name)
    np.savez_compressed(path, recs=out, width=np.array(w), height=np.array(h))
    print("%-24s %d pictures (%d of %d records) -> %s (%.0f KiB); mm_sao %s layers %s luma types %s chroma types %s merges L%d U%d" %
          (name, len(chosen), len(out), len(recs), os.path.basename(path), os.path.getsize(path) / 1024, np.unique(out["mm_sao"]).tolist(),
           np.unique(out["temporal_layer"]).tolist(), np.bincount(out["out"]["type"][:, 0], minlength=6).tolist(),
           np.bincount(out["out"]["type"][:, 1], minlength=6).tolist(), int(out["out"]["merge_left"].sum()), int(out["out"]["merge_up"].sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
