#!/usr/bin/env python3
"""Generate the TU-reconstruction golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_RECON_DUMP set, so the table-slot interposer of
oracle/ref_harness_recon_dump.c records a sample of the EncodeGenerateRecon(16bit) calls of the real encode pass: per
reconstructed plane the inverse-quantised coefficients, the prediction and the reconstruction.
Stored as tests/golden/recon_<name>.npz.  Needs /root/reference (this container only).
Usage: python tests/golden/make_recon_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

REC = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("size", "<u4"), ("plane", "<u4"), ("only_dc", "<u4"), ("dst", "<u4"),
                ("bytes_per_sample", "<u4"), ("pad", "<u4"), ("coeff", "<i2", 1024), ("pred", "<u2", 1024), ("recon", "<u2", 1024)])

# name -> (clip kind, width, height, frames, seed, bit depth, encoder args, sampling stride, records kept)
CASES = {
    "ip_416x240_m9": ("motion", 416, 240, 4, 7, 8, ["-encMode", "9", "-pred-struct", "0"], 1, 320),
    "i_noise_192x128_m1": ("noise", 192, 128, 2, 11, 8, ["-encMode", "1", "-intra-period", "0", "-q", "25"], 9, 320),
    "b10_416x240_m7": ("motion", 416, 240, 6, 7, 10, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-bit-depth", "10"], 2, 320),
    "noise_320x256_m6_q22": ("noise", 320, 256, 3, 11, 8, ["-encMode", "6", "-pred-struct", "1", "-q", "22"], 5, 320),
}


def run_case(name):
    kind, w, h, n, seed, depth, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "recon.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_RECON_DUMP=dump, SVT_REF_RECON_STRIDE=str(stride)), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    order = np.lexsort((recs["only_dc"], recs["plane"], recs["size"]))
    sel = order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]
    recs = recs[np.sort(np.unique(sel))]
    out = {k: recs[k] for k in ("size", "plane", "only_dc", "dst", "bytes_per_sample")}
    for k in ("coeff", "pred", "recon"):
        out[k] = np.concatenate([r[k][: int(r["size"]) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "recon_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-24s %d records (sizes %s) -> %s (%.0f KiB); planes %s, only-dc %d, dst %d, bps %s" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["plane"]).tolist(), int(recs["only_dc"].sum()), int(recs["dst"].sum()),
           np.unique(recs["bytes_per_sample"]).tolist()))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
