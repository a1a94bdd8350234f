#!/usr/bin/env python3
"""Generate the ME golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref (the reference compiled from /root/reference by
oracle/Makefile) on seeded synthetic clips with SVT_REF_ME_DUMP set, so that the
--wrap interposer oracle/ref_harness_me_dump.c records, for every LCU of every
inter picture, the controls MotionEstimateLcu read and everything it produced.
The records are stored as tests/golden/me_<name>.npz.

Needs /root/reference (this container only); the fixtures travel, this script's
inputs do not.  Usage:  python tests/golden/make_me_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

# name -> (clip kind, width, height, frames, seed, encoder args, max pictures kept)
CASES = {
    # 576p class: HME L0+L1+L2, no search-centre update, P pictures
    "p_640x384_m9": ("motion", 640, 384, 4, 7, ["-encMode", "9", "-pred-struct", "0"], 3),
    # 1080i class: search-centre update, B pictures (bi-pred, list-1 direct centre, equal-POC refs)
    "b_1024x768_m7": ("motion", 1024, 768, 5, 7,
                      ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 4),
    # BASELINE config 2: 1080p, encMode 9, low-delay P (partial bottom LCU row)
    "p_1920x1080_m9": ("motion", 1920, 1080, 2, 7, ["-encMode", "9", "-pred-struct", "0"], 1),
    # tie-break rules: every SAD ties on a flat clip
    "flat_320x256_m9": ("flat", 320, 256, 3, 5, ["-encMode", "9", "-pred-struct", "1"], 2),
    # SSD sub-pel search on all PUs (encMode 4), worst-case residual energy
    "noise_320x256_m4": ("noise", 320, 256, 3, 11, ["-encMode", "4"], 2),
    # encMode 1: 64x64 fractional search, 64x64 search area, cu8x8 refinement
    "p_320x256_m1": ("motion", 320, 256, 3, 7, ["-encMode", "1"], 2),
    # BASELINE config 3 (4K, encMode 7, random access, 2 hierarchical levels, 60 fps): one B picture, four LCU rows
    # kept (top, two interior, the partial bottom row) to bound the fixture size
    "b_3840x2160_m7": ("motion", 3840, 2160, 5, 7,
                       ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60"], 1),
}
ROWS_KEPT = {"b_3840x2160_m7": [0, 11, 22, 33]}


def run_case(name):
    kind, w, h, n, seed, args, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "clip.yuv")
        dump = os.path.join(td, "me.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        env = dict(os.environ, SVT_REF_ME_DUMP=dump)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32",
               "-asm", "0", "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=S.DUMP_DTYPE)
    assert len(recs) and (recs["record_size"] == S.DUMP_DTYPE.itemsize).all()
    nl = S.lcu_count(w, h)
    pics = sorted(set(int(p) for p in recs["picture_number"]))[:keep]
    out = {"clip": np.array([kind, str(w), str(h), str(n), str(seed)]), "enc_args": np.array(args)}
    meta, params, results = [], [], []
    for pn in pics:
        rr = recs[recs["picture_number"] == pn]
        rr = rr[np.argsort(rr["lcu_index"])]
        assert len(rr) == nl, (name, pn, len(rr), nl)
        r0 = rr[0]
        # the padded input the reference searched IS the raw clip frame (no denoising at these settings)
        assert S.plane_checksum(S.gen_luma(kind, w, h, pn, seed)) == int(r0["luma_crc"]), "input mismatch"
        meta.append([pn, int(r0["slice_type"]), int(r0["ref_poc"][0]), int(r0["ref_poc"][1])])
        params.append(r0["params"])
        assert all(rr["params"] == r0["params"])
        res = rr["result"].copy()
        if name in ROWS_KEPT:  # zero the rows that are not kept (they then compress to nothing)
            wl = (w + 63) // 64
            keep_mask = np.zeros(nl, bool)
            for r in ROWS_KEPT[name]:
                keep_mask[r * wl:(r + 1) * wl] = True
            res[~keep_mask] = np.zeros((), res.dtype)
        results.append(res)
    out["meta"] = np.array(meta, np.int64)           # picture, slice type, ref POC l0, ref POC l1
    out["params"] = np.array(params, S.ME_PARAMS_DTYPE)
    out["results"] = np.stack(results)                # [picture][lcu] ME_LCU_DTYPE
    if name in ROWS_KEPT:
        out["rows_kept"] = np.array(ROWS_KEPT[name])
    path = os.path.join(S.GOLDEN_DIR, "me_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-20s %d pictures x %d LCUs -> %s (%.0f KiB)" % (name, len(pics), nl, os.path.basename(path),
                                                           os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
