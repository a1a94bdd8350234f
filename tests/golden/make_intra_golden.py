#!/usr/bin/env python3
"""Generate the encode-pass intra-prediction golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_INTRA_DUMP set, so the table-slot interposers of
oracle/ref_harness_intra_dump.c record a sample of the GenerateIntraReferenceSamplesEncodePass + EncodePassIntraPrediction
call pairs of the real encode pass: neighbour-array slices and flags in, the three predicted blocks out.
Stored as tests/golden/intra_<name>.npz.  Needs /root/reference (this container only).
With SVT_REF_INTRA_MD_DUMP the same harness records the mode decision's closed-loop IntraPredictionCl calls (one record per
luma block, one per chroma pair; component_mask says which planes are valid) -> tests/golden/intramd_<name>.npz.
Usage: python tests/golden/make_intra_golden.py [name ...]      (names of either family)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

REC = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("size", "<u4"), ("bytes_per_sample", "<u4"),
                ("constrained_intra", "u1"), ("strong_smoothing", "u1"), ("pic_left", "u1"), ("pic_top", "u1"), ("pic_right", "u1"),
                ("bottom_left_ok", "u1"), ("top_right_ok", "u1"), ("pad0", "u1"), ("luma_mode", "<u4"), ("chroma_mode", "<u4"),
                ("component_mask", "<u4"), ("pad1", "<u4"), ("mode_left", "u1", 32), ("mode_top", "u1", 32), ("mode_tl", "u1"),
                ("pad2", "u1", 7), ("left", "<u2", (3, 128)), ("top", "<u2", (3, 128)), ("tl", "<u2", 3), ("pad3", "<u2"),
                ("pred_y", "<u2", 4096), ("pred_cb", "<u2", 1024), ("pred_cr", "<u2", 1024)])

# name -> (clip kind, width, height, frames, seed, bit depth, encoder args, sampling stride, records kept)
CASES = {
    "i_416x240_m9": ("motion", 416, 240, 2, 7, 8, ["-encMode", "9", "-intra-period", "0"], 1, 260),
    "i_noise_200x136_m5": ("noise", 200, 136, 2, 11, 8, ["-encMode", "5", "-intra-period", "0", "-q", "28"], 3, 260),
    "p_noise_320x256_m6": ("noise", 320, 256, 3, 11, 8, ["-encMode", "6", "-pred-struct", "0", "-q", "25"], 2, 200),
    "i10_416x240_m7": ("motion", 416, 240, 2, 7, 10, ["-encMode", "7", "-intra-period", "0", "-bit-depth", "10"], 1, 200),
    "i_tiles_640x384_m9": ("motion", 640, 384, 2, 7, 8, ["-encMode", "9", "-intra-period", "0", "-tile_row_cnt", "2", "-tile_col_cnt", "2"], 2, 200),
    "p_cip_noise_320x256_m6": ("noise", 320, 256, 4, 11, 8, ["-encMode", "6", "-pred-struct", "0", "-q", "25", "-constrd-intra", "1"], 2, 200),
}
# mode-decision side: name -> (clip kind, width, height, frames, seed, encoder args, sampling stride, records kept)
MD_CASES = {
    "i_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0"], 5, 260),
    "ip_noise_320x256_m1": ("noise", 320, 256, 3, 11, ["-encMode", "1", "-pred-struct", "0", "-q", "28"], 61, 260),
    "p_motion_416x240_m5": ("motion", 416, 240, 4, 7, ["-encMode", "5", "-pred-struct", "0"], 11, 220),
    "i_tiles_640x384_m3": ("motion", 640, 384, 2, 7, ["-encMode", "3", "-intra-period", "0", "-tile_row_cnt", "2", "-tile_col_cnt", "2"], 23, 260),
}
# open-loop mode decision (IntraPredictionOl: non-reference / upper-layer pictures): same fields, neighbours = source samples
OL_CASES = {
    "b_416x240_m7": ("motion", 416, 240, 9, 7, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 7, 240),
    "p_noise_200x136_m9": ("noise", 200, 136, 5, 11, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "1", "-q", "28"], 1, 240),
    "b_noise_200x136_m4": ("noise", 200, 136, 9, 11, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "26"], 13, 300),
}
# intra 4x4 coding units of the encode pass (encMode <= 2): a 4x4 luma partition (size 4, component_mask 1) or the chroma pair of
# its 8x8 coding unit (size 8, component_mask 6; the predicted chroma blocks are 4x4)
I4_CASES = {
    "i_noise_200x136_m1": ("noise", 200, 136, 2, 11, 8, ["-encMode", "1", "-intra-period", "0", "-q", "24"], 3, 300),
    "i10_motion_416x240_m2": ("motion", 416, 240, 1, 7, 10, ["-encMode", "2", "-intra-period", "0", "-q", "22", "-bit-depth", "10"], 3, 240),
    # the mode decision's 4x4 search (Intra4x4IntraPredictionCl): names starting with "md_"
    "md_i_noise_200x136_m1": ("noise", 200, 136, 1, 11, 8, ["-encMode", "1", "-intra-period", "0", "-q", "24"], 97, 300),
}
KEEP = ("size", "bytes_per_sample", "constrained_intra", "strong_smoothing", "pic_left", "pic_top", "pic_right", "bottom_left_ok",
        "top_right_ok", "luma_mode", "chroma_mode", "mode_left", "mode_top", "mode_tl", "left", "top", "tl")


def run_case(name):
    kind, w, h, n, seed, depth, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "intra.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_INTRA_DUMP=dump, SVT_REF_INTRA_STRIDE=str(stride)), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    order = np.lexsort((recs["luma_mode"], recs["size"]))
    sel = order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]
    recs = recs[np.sort(np.unique(sel))]
    out = {k: recs[k] for k in KEEP}
    out["pred_y"] = np.concatenate([r["pred_y"][: int(r["size"]) ** 2] for r in recs])
    out["pred_cb"] = np.concatenate([r["pred_cb"][: (int(r["size"]) // 2) ** 2] for r in recs])
    out["pred_cr"] = np.concatenate([r["pred_cr"][: (int(r["size"]) // 2) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "intra_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-26s %d records (sizes %s) -> %s (%.0f KiB); %d luma modes, edges L/T/R %d/%d/%d, bl/tr ok %d/%d, inter neighbours %d, cip %d" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           len(np.unique(recs["luma_mode"])), int(recs["pic_left"].sum()), int(recs["pic_top"].sum()), int(recs["pic_right"].sum()),
           int(recs["bottom_left_ok"].sum()), int(recs["top_right_ok"].sum()),
           int(((recs["mode_left"] == 1).any(axis=1) | (recs["mode_top"] == 1).any(axis=1)).sum()), int(recs["constrained_intra"].sum())))


def run_i4_case(name):
    kind, w, h, n, seed, depth, args, stride, keep = I4_CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "intra4.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + args
        env = dict(os.environ, SVT_REF_INTRA4_DUMP=dump, SVT_REF_INTRA4_STRIDE=str(stride))
        if name.startswith("md_"):
            env["SVT_REF_INTRA4_MD"] = "1"
        subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    total = len(recs)
    groups = [np.flatnonzero(recs["component_mask"] == m) for m in np.unique(recs["component_mask"])]
    groups = [g[np.argsort(recs["luma_mode"][g], kind="stable")] for g in groups if len(g)]
    share = keep // len(groups)
    sel = np.concatenate([g[np.linspace(0, len(g) - 1, min(share, len(g))).astype(int)] for g in groups])
    recs = recs[np.sort(np.unique(sel))]
    out = {k: recs[k] for k in KEEP + ("component_mask",)}
    out["pred_y"] = np.concatenate([r["pred_y"][: int(r["size"]) ** 2] for r in recs])
    out["pred_cb"] = np.concatenate([r["pred_cb"][: (int(r["size"]) // 2) ** 2] for r in recs])
    out["pred_cr"] = np.concatenate([r["pred_cr"][: (int(r["size"]) // 2) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "intra4_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-26s %d of %d records (luma 4x4 %d / chroma pairs %d) -> %s (%.0f KiB); %d luma modes, edges L/T/R %d/%d/%d, bl/tr ok %d/%d" %
          (name, len(recs), total, int((recs["component_mask"] == 1).sum()), int((recs["component_mask"] != 1).sum()),
           os.path.basename(path), os.path.getsize(path) / 1024, len(np.unique(recs["luma_mode"])), int(recs["pic_left"].sum()),
           int(recs["pic_top"].sum()), int(recs["pic_right"].sum()), int(recs["bottom_left_ok"].sum()), int(recs["top_right_ok"].sum())))


def run_md_case(name, ol=False):
    kind, w, h, n, seed, args, stride, keep = (OL_CASES if ol else MD_CASES)[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "intramd.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + \
            ([] if "-q" in args else ["-q", "32"]) + args
        env = dict(os.environ, SVT_REF_INTRA_MD_DUMP=dump, SVT_REF_INTRA_MD_STRIDE=str(stride))
        if ol:
            env["SVT_REF_INTRA_MD_OL"] = "1"
        subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    total = len(recs)
    groups = [np.flatnonzero((recs["component_mask"] == m) & (recs["size"] == z)) for m in np.unique(recs["component_mask"])
              for z in np.unique(recs["size"])]
    groups = [g[np.argsort(recs["luma_mode"][g], kind="stable")] for g in groups if len(g)]
    share = keep // len(groups)
    sel = np.concatenate([g[np.linspace(0, len(g) - 1, min(share, len(g))).astype(int)] for g in groups])
    recs = recs[np.sort(np.unique(sel))]
    out = {k: recs[k] for k in KEEP + ("component_mask",)}
    out["no_smoothing"] = recs["pad0"]
    out["pred_y"] = np.concatenate([r["pred_y"][: int(r["size"]) ** 2] for r in recs])
    out["pred_cb"] = np.concatenate([r["pred_cb"][: (int(r["size"]) // 2) ** 2] for r in recs])
    out["pred_cr"] = np.concatenate([r["pred_cr"][: (int(r["size"]) // 2) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, ("intraol_%s.npz" if ol else "intramd_%s.npz") % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-26s %d of %d records (sizes %s, luma %d / chroma %d) -> %s (%.0f KiB); %d luma modes, edges L/T/R %d/%d/%d, inter neighbours %d" %
          (name, len(recs), total, dict(zip(sizes.tolist(), cnt.tolist())), int((recs["component_mask"] == 1).sum()),
           int((recs["component_mask"] != 1).sum()), os.path.basename(path), os.path.getsize(path) / 1024,
           len(np.unique(recs["luma_mode"])), int(recs["pic_left"].sum()), int(recs["pic_top"].sum()), int(recs["pic_right"].sum()),
           int(((recs["mode_left"] == 1).any(axis=1) | (recs["mode_top"] == 1).any(axis=1)).sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    names = sys.argv[1:] or (list(CASES) + ["md:" + k for k in MD_CASES] + ["ol:" + k for k in OL_CASES] + ["i4:" + k for k in I4_CASES])
    for nm in names:
        if nm.startswith("md:"):
            run_md_case(nm[3:])
        elif nm.startswith("ol:"):
            run_md_case(nm[3:], ol=True)
        elif nm.startswith("i4:"):
            run_i4_case(nm[3:])
        else:
            run_case(nm)
