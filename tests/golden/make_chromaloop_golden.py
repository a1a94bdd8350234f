#!/usr/bin/env python3
"""Generate the chroma full-loop golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_CHROMALOOP_DUMP set, so the --wrap interposers of
oracle/ref_harness_chromaloop_dump.c record a sample of the FullLoop_R + CuFullDistortionFastTuMode_R call pairs (inputs
and outputs) made inside the real mode decision.  Stored as tests/golden/chromaloop_<name>.npz (only the (size/2)^2 part of
each 32x32 plane array).  Needs /root/reference (this container only).
Usage: python tests/golden/make_chromaloop_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

COST = np.dtype([("last", "<u4", 176), ("sig", "u1", 84), ("g1", "u1", 48), ("g2", "u1", 12), ("sigml", "u1", 8),
                 ("g1x", "<u2", 96), ("sigv", "u1", (32, 16))])
REC = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("size", "<u4"), ("origin_x", "<u4"),
                ("origin_y", "<u4"), ("cb_qp", "<u4"), ("cr_qp", "<u4"), ("slice_type", "<u4"), ("temporal_layer", "<u4"),
                ("pf_mode", "<u4"), ("cand_type", "<u4"), ("intra_luma_mode", "<u4"), ("cbf_in", "<u4", 2),
                ("cbf_out", "<u4", 2), ("nz_out", "<u4", (2, 5)), ("bits_in", "<u8", 2), ("bits_out", "<u8", 2),
                ("dist_in", "<u8", (2, 2)), ("dist_out", "<u8", (2, 2)), ("cost", COST), ("residual", "<i2", (2, 1024)),
                ("quant", "<i2", (2, 1024)), ("recon", "<i2", (2, 1024))], align=True)

# name -> (clip kind, width, height, frames, seed, encoder args, sampling stride, records kept)
CASES = {
    "p_416x240_m5": ("motion", 416, 240, 4, 7, ["-encMode", "5", "-pred-struct", "0"], 37, 300),
    "p_416x240_m9": ("motion", 416, 240, 5, 7, ["-encMode", "9", "-pred-struct", "0"], 5, 240),
    "b_416x240_m7": ("motion", 416, 240, 10, 7, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 17, 260),
    "noise_320x256_m6": ("noise", 320, 256, 3, 11, ["-encMode", "6", "-pred-struct", "1", "-q", "22"], 13, 240),
    # encMode 4 P pictures: rdoqPmCoreMethod == EB_PMCORE, which re-decides luma blocks only - the chroma pair is unchanged
    "pm_p_416x240_m4": ("motion", 416, 240, 4, 7, ["-encMode", "4", "-pred-struct", "0"], 41, 240),
}


def run_case(name):
    kind, w, h, n, seed, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "cl.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_CHROMALOOP_DUMP=dump, SVT_REF_CHROMALOOP_STRIDE=str(stride)),
                       check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    order = np.lexsort((recs["picture_number"], recs["size"]))
    sel = order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]
    recs = recs[np.sort(sel)]
    out = {k: recs[k] for k in REC.names if k not in ("residual", "quant", "recon", "magic", "record_size")}
    for k in ("residual", "quant", "recon"):  # pack: only (size/2)^2 samples per plane are meaningful
        out[k] = np.concatenate([r[k][:, : (int(r["size"]) // 2) ** 2].reshape(-1) for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "chromaloop_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-28s %d records (sizes %s) -> %s (%.0f KiB); types %s, pf %s, slices %s, nonzero-cbf %d" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["cand_type"]).tolist(), np.unique(recs["pf_mode"]).tolist(), np.unique(recs["slice_type"]).tolist(),
           int((recs["cbf_out"] != 0).any(axis=1).sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
