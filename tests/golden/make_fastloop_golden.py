#!/usr/bin/env python3
"""Generate the fast-loop distortion golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_FASTLOOP_DUMP set, so the --wrap interposers of
oracle/ref_harness_fastloop_dump.c record a sample of the second-loop iterations of ProductPerformFastLoop: the source block,
the candidate's predicted block (three planes), the luma / chroma distortions the loop handed to the fast-cost function and
the switches that shape them.  Stored as tests/golden/fastloop_<name>.npz.  Needs /root/reference (this container only).
Usage: python tests/golden/make_fastloop_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

SCALARS = ("size", "cand_type", "slice_type", "use_chroma", "mpm_flag", "distortion_ready", "noise_lcu", "intra_luma_mode")
REC = np.dtype([("magic", "<u4"), ("record_size", "<u4")] + [(k, "<u4") for k in SCALARS] +
               [("luma_distortion", "<u8"), ("chroma_distortion", "<u8"), ("me_distortion", "<u8"),
                ("src_y", "u1", 4096), ("src_cb", "u1", 1024), ("src_cr", "u1", 1024),
                ("pred_y", "u1", 4096), ("pred_cb", "u1", 1024), ("pred_cr", "u1", 1024)], align=True)

# name -> (clip kind, width, height, frames, seed, encoder args, sampling stride, records kept)
CASES = {
    "p_416x240_m9": ("motion", 416, 240, 4, 7, ["-encMode", "9", "-pred-struct", "0"], 7, 240),
    "b_416x240_m5": ("motion", 416, 240, 9, 7, ["-encMode", "5", "-pred-struct", "2", "-hierarchical-levels", "2"], 61, 260),
    "ip_noise_320x256_m1": ("noise", 320, 256, 3, 11, ["-encMode", "1", "-pred-struct", "0", "-q", "28"], 211, 260),
    "i_416x240_m7": ("motion", 416, 240, 2, 7, ["-encMode", "7", "-intra-period", "0"], 23, 220),
}


def run_case(name):
    kind, w, h, n, seed, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "fastloop.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + \
            ([] if "-q" in args else ["-q", "32"]) + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_FASTLOOP_DUMP=dump, SVT_REF_FASTLOOP_STRIDE=str(stride)), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    total = len(recs)
    groups = [np.flatnonzero((recs["size"] == z) & (recs["cand_type"] == t)) for z in np.unique(recs["size"]) for t in (1, 2)]
    groups = [g for g in groups if len(g)]
    share = keep // len(groups)
    sel = np.sort(np.concatenate([g[np.linspace(0, len(g) - 1, min(share, len(g))).astype(int)] for g in groups]))
    recs = recs[np.unique(sel)]
    out = {k: recs[k] for k in SCALARS + ("luma_distortion", "chroma_distortion", "me_distortion")}
    for k in ("src_y", "pred_y"):
        out[k] = np.concatenate([r[k][: int(r["size"]) ** 2] for r in recs])
    for k in ("src_cb", "src_cr", "pred_cb", "pred_cr"):
        out[k] = np.concatenate([r[k][: (int(r["size"]) // 2) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "fastloop_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-22s %d of %d records (sizes %s) -> %s (%.0f KiB); types %s, chroma in the loop %s, mpm %d, noise LCUs %d" %
          (name, len(recs), total, dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["cand_type"]).tolist(), np.unique(recs["use_chroma"]).tolist(), int(recs["mpm_flag"].sum()),
           int(recs["noise_lcu"].sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
