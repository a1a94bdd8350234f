#!/usr/bin/env python3
"""Generate the encode-pass quantiser golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_UQIQ_DUMP set, so the --wrap interposer of
oracle/ref_harness_uqiq_dump.c records a sample of the UnifiedQuantizeInvQuantize calls of the real encode pass (no RDOQ, no
perceptual masking): coefficients and scalars in, quantised / reconstructed coefficients and the non-zero count out.
Stored as tests/golden/uqiq_<name>.npz.  Needs /root/reference (this container only).
With SVT_REF_UQIQPM_DUMP the same interposer records the calls of encMode 1..4 encodes, which run the PM-core variant
(rdoqPmCoreMethod == EB_PMCORE), together with the lambda and the CabacCost_t tables they price with -> uqiqpm_<name>.npz
(names starting with "pm:").
Usage: python tests/golden/make_uqiq_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

SCALARS = ("size", "qp", "bit_depth", "slice_type", "shape", "clean_sparse", "enable_cb_flag", "contouring_flag", "component",
           "temporal_layer", "dz_offset", "nz_out")
REC = np.dtype([("magic", "<u4"), ("record_size", "<u4")] + [(k, "<u4") for k in SCALARS] +
               [(k, "<i2", 1024) for k in ("coeff", "quant_in", "recon_in", "quant", "recon")])

# name -> (clip kind, width, height, frames, seed, bit depth, encoder args, sampling stride, records kept)
CASES = {
    "ip_416x240_m9": ("motion", 416, 240, 4, 7, 8, ["-encMode", "9", "-pred-struct", "0"], 3, 320),
    "b_416x240_m7": ("motion", 416, 240, 9, 7, 8, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 5, 320),
    "noise_320x256_m6": ("noise", 320, 256, 3, 11, 8, ["-encMode", "6", "-pred-struct", "1"], 23, 320),
    "i10_416x240_m7": ("motion", 416, 240, 2, 7, 10, ["-encMode", "7", "-intra-period", "0", "-bit-depth", "10"], 3, 260),
    "vbr_416x240_m5": ("motion", 416, 240, 6, 7, 8, ["-encMode", "5", "-rc", "1", "-tbr", "300000"], 7, 320),
}


COST = np.dtype([("last", "<u4", 176), ("sig", "u1", 84), ("g1", "u1", 48), ("g2", "u1", 12), ("sigml", "u1", 8),
                 ("g1x", "<u2", 96), ("sigv", "u1", (32, 16))])
PM_SCALARS = ("size", "qp", "bit_depth", "slice_type", "component", "cand_type", "lambda", "nz_out")
PM_REC = np.dtype([("magic", "<u4"), ("record_size", "<u4")] + [(k, "<u4") for k in PM_SCALARS] + [("cost", COST)] +
                  [(k, "<i2", 1024) for k in ("coeff", "quant", "recon")], align=True)
PM_CASES = {
    "p_416x240_m4": ("motion", 416, 240, 4, 7, 8, ["-encMode", "4", "-pred-struct", "0"], 3, 300),
    "b_noise_320x256_m2": ("noise", 320, 256, 5, 11, 8, ["-encMode", "2", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "30"], 11, 300),
    "i10_416x240_m3": ("motion", 416, 240, 2, 7, 10, ["-encMode", "3", "-intra-period", "0", "-bit-depth", "10", "-q", "28"], 5, 260),
}


def run_pm_case(name):
    kind, w, h, n, seed, depth, args, stride, keep = PM_CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "uqiqpm.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + \
            ([] if "-q" in args else ["-q", "32"]) + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_UQIQPM_DUMP=dump, SVT_REF_UQIQPM_STRIDE=str(stride)), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=PM_REC)
    assert len(recs) and (recs["record_size"] == PM_REC.itemsize).all(), (len(recs), PM_REC.itemsize)
    order = np.lexsort((recs["nz_out"], recs["component"], recs["size"]))
    sel = np.sort(np.unique(order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]))
    recs = recs[sel]
    out = {k: recs[k] for k in PM_SCALARS}
    # the rate tables are per picture: store the distinct ones once
    keys = [r["cost"].tobytes() for r in recs]
    uniq = sorted(set(keys))
    out["cost_tables"] = np.array([np.frombuffer(k, COST)[0] for k in uniq], COST)
    out["cost_index"] = np.array([uniq.index(k) for k in keys], np.int32)
    for k in ("coeff", "quant", "recon"):
        out[k] = np.concatenate([r[k][: int(r["size"]) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "uqiqpm_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-20s %d records (sizes %s) -> %s (%.0f KiB); components %s, types %s, slices %s, depth %s, zero %d, %d rate tables" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["component"]).tolist(), np.unique(recs["cand_type"]).tolist(), np.unique(recs["slice_type"]).tolist(),
           np.unique(recs["bit_depth"]).tolist(), int((recs["nz_out"] == 0).sum()), len(uniq)))


def run_case(name):
    kind, w, h, n, seed, depth, args, stride, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "uqiq.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_UQIQ_DUMP=dump, SVT_REF_UQIQ_STRIDE=str(stride)), check=True,
                       stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=REC)
    assert len(recs) and (recs["record_size"] == REC.itemsize).all(), (len(recs), REC.itemsize)
    order = np.lexsort((recs["nz_out"], recs["component"], recs["shape"], recs["size"]))
    sel = np.sort(np.unique(order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]))
    recs = recs[sel]
    out = {k: recs[k] for k in SCALARS}
    for k in ("coeff", "quant_in", "recon_in", "quant", "recon"):
        out[k] = np.concatenate([r[k][: int(r["size"]) ** 2] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "uqiq_%s.npz" % name)
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(recs["size"], return_counts=True)
    print("%-20s %d records (sizes %s) -> %s (%.0f KiB); shapes %s, clean-sparse %d, cb-flag %d, contouring %d, dz %s, slices %s, depth %s, zero %d" %
          (name, len(recs), dict(zip(sizes.tolist(), cnt.tolist())), os.path.basename(path), os.path.getsize(path) / 1024,
           np.unique(recs["shape"]).tolist(), int(recs["clean_sparse"].sum()), int(recs["enable_cb_flag"].sum()),
           int(recs["contouring_flag"].sum()), np.unique(recs["dz_offset"]).tolist(), np.unique(recs["slice_type"]).tolist(),
           np.unique(recs["bit_depth"]).tolist(), int((recs["nz_out"] == 0).sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or (list(CASES) + ["pm:" + k for k in PM_CASES])):
        if nm.startswith("pm:"):
            run_pm_case(nm[3:])
        else:
            run_case(nm)
