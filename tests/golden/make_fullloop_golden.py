#!/usr/bin/env python3
"""Generate the luma full-loop golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_FULLLOOP_DUMP set, so the --wrap interposer
oracle/ref_harness_fullloop_dump.c records a sample of the ProductFullLoop calls (inputs and outputs) made inside the
real mode decision.  Stored as tests/golden/fullloop_<name>.npz (only the size x size part of each 64x64 array).
Needs /root/reference (this container only).  Usage: python tests/golden/make_fullloop_golden.py [name ...]
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
                ("origin_y", "<u4"), ("qp", "<u4"), ("slice_type", "<u4"), ("temporal_layer", "<u4"), ("pf_mode", "<u4"),
                ("cand_type", "<u4"), ("intra_luma_mode", "<u4"), ("full_lambda", "<u4"), ("cbf_bits", "<u4", 4),
                ("ycbf_before", "<u4"), ("ycbf_after", "<u4"), ("nz_in", "<u4", 5), ("nz_out", "<u4", 5),
                ("bits_in", "<u8"), ("bits_out", "<u8"), ("dist_in", "<u8", 2), ("dist_out", "<u8", 2),
                ("ydc", "<i2", 4), ("cand_nz", "<u2", 4), ("cost", COST), ("residual", "<i2", 4096),
                ("quant", "<i2", 4096), ("recon", "<i2", 4096), ("cabac_update", "<u4"), ("pad", "<u4"), ("ctx_in", "<u4", 136),
                ("ctx_out", "<u4", 136)], align=True)

# name -> (clip kind, width, height, frames, seed, encoder args, sampling stride, records kept)
CASES = {
    "ip_416x240_m9": ("motion", 416, 240, 4, 7, ["-encMode", "9", "-pred-struct", "0"], 29, 260),
    "i_1920x1080_m10": ("motion", 1920, 1080, 1, 7, ["-encMode", "10", "-intra-period", "0"], 211, 200),
    "b_416x240_m7": ("motion", 416, 240, 6, 7, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 23, 260),
    "noise_320x256_m6": ("noise", 320, 256, 3, 11, ["-encMode", "6", "-pred-struct", "1", "-q", "22"], 31, 200),
    # encMode 1..4: PM-core re-decision of every 4x4 block of levels (contextPtr->rdoqPmCoreMethod == EB_PMCORE).  Only the P / B
    # pictures of encMode 4 run it without the CABAC-context-updating rate estimator (coeffCabacUpdate, EbEncDecProcess.c:2116)
    "pm_b_noise_320x256_m4": ("noise", 320, 256, 5, 11, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "27"], 89, 260),
    "pm_p_motion_416x240_m4": ("motion", 416, 240, 4, 7, ["-encMode", "4", "-pred-struct", "0", "-q", "30"], 61, 260),
    # coeffCabacUpdate (EbEncDecProcess.c:2115-2123): the I picture of every clip below 4K at encMode <= 9 decides over the full depth
    # range with chroma in the loop, and prices coefficients with the CABAC-context-UPDATING estimator; the records carry the
    # candidate's context model before and after the call.  encMode 7: plain quantiser; encMode 3: PM-core on every picture
    "cabac_i_motion_416x240_m7": ("motion", 416, 240, 1, 7, ["-encMode", "7", "-q", "30"], 37, 300),
    "cabac_i_noise_320x256_m7": ("noise", 320, 256, 1, 11, ["-encMode", "7", "-q", "36"], 41, 220),
    "cabac_pm_ib_motion_416x240_m3": ("motion", 416, 240, 5, 7, ["-encMode", "3", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "30"], 211, 300),
}


def run_case(name):
    kind, w, h, n, seed, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "fl.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + \
            ([] if "-q" in args else ["-q", "32"]) + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_FULLLOOP_DUMP=dump, SVT_REF_FULLLOOP_STRIDE=str(stride)),
                       check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    # keep a spread of sizes / types
    order = np.lexsort((recs["picture_number"], recs["size"]))
    sel = order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]
    recs = recs[np.sort(sel)]
    out = {k: recs[k] for k in REC.names if k not in ("residual", "quant", "recon", "magic", "record_size", "pad", "ctx_in", "ctx_out")}
    if recs["cabac_update"].any():
        assert recs["cabac_update"].all() or name.startswith("cabac_pm"), "mixed fixture"
        out["ctx_in"], out["ctx_out"] = recs["ctx_in"], recs["ctx_out"]
    else:
        del out["cabac_update"]
    out["pm_core"], out["pf_mode"] = recs["pf_mode"] >> 16, recs["pf_mode"] & 0xffff   # the harness packs both into one word
    for k in ("residual", "quant", "recon"):  # pack: only size*size samples are meaningful
        out[k] = np.concatenate([r[k][: int(r["size"]) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "fullloop_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-28s %d records (sizes %s) -> %s (%.0f KiB); types %s, pf %s, pm-core %s, slices %s" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["cand_type"]).tolist(), np.unique(out["pf_mode"]).tolist(), np.unique(out["pm_core"]).tolist(), np.unique(recs["slice_type"]).tolist()))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
