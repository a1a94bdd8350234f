#!/usr/bin/env python3
"""Generate the open-loop intra search (OIS) golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded synthetic clips with SVT_REF_OIS_DUMP set, so the --wrap interposer
oracle/ref_harness_ois_dump.c records, per LCU, the controls OpenLoopIntraSearchLcu read, the ME distortions it
consulted and its result arrays before and after the call.  Stored as tests/golden/ois_<name>.npz.
Needs /root/reference (this container only).  Usage: python tests/golden/make_ois_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

# name -> (clip kind, width, height, frames, seed, encoder args, pictures kept)
CASES = {
    # I picture with 8x8 OIS + P pictures (OIS points, stage-1 search), partial right/bottom LCUs
    "ip_416x240_m9": ("motion", 416, 240, 4, 7, ["-encMode", "9", "-pred-struct", "0"], 3),
    # BASELINE config 1 class (1080p encMode 10, the only class where it is legal): P pictures limited to DC
    # (OpenLoopIntraDC), I picture with 8x8 OIS
    "ip_1920x1080_m10": ("motion", 1920, 1080, 2, 7, ["-encMode", "10", "-pred-struct", "0"], 2),
    # encMode 4 (< 4K): oisKernelLevel on temporal layer 0 (all 35 modes, best 18 sorted), B pictures, 8x8 CUs
    "ib_320x256_m4": ("motion", 320, 256, 5, 7, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2"], 5),
    # same preset, flat low-delay P: every picture is temporal layer 0 -> 35-mode search on P pictures
    "ip_320x256_m4_flat": ("motion", 320, 256, 3, 7, ["-encMode", "4", "-pred-struct", "0", "-hierarchical-levels", "0"], 3),
    # ties and zero SADs everywhere
    "flat_320x256_m9": ("flat", 320, 256, 3, 5, ["-encMode", "9", "-pred-struct", "1"], 3),
    # worst-case residual energy, heavy threshold set (encMode 6, < 4K)
    "noise_320x256_m6": ("noise", 320, 256, 3, 11, ["-encMode", "6", "-pred-struct", "1"], 3),
    # BASELINE config 2: one I and one P picture of 1080p encMode 9 (partial bottom LCU row)
    "ip_1920x1080_m9": ("motion", 1920, 1080, 2, 7, ["-encMode", "9", "-pred-struct", "0"], 2),
    # BASELINE configs[2] (4K, encMode 7, random access, 2 hierarchical levels, 60 fps): the I picture and the B pictures of
    # temporal layers 0 / 1 / 2; three LCU rows kept (top, interior, the partial bottom row) to bound the fixture size
    "ib_3840x2160_m7": ("motion", 3840, 2160, 5, 7,
                        ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60"], 5),
}
ROWS_KEPT = {"ib_3840x2160_m7": [0, 16, 33]}


def run_case(name):
    kind, w, h, n, seed, args, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "ois.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_OIS_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=S.OIS_DUMP_DTYPE)
    assert len(recs) and (recs["record_size"] == S.OIS_DUMP_DTYPE.itemsize).all()
    nl = S.lcu_count(w, h)
    pics = sorted(set(int(p) for p in recs["picture_number"]))[:keep]
    meta, params, me_sad, before, after = [], [], [], [], []
    for pn in pics:
        rr = recs[recs["picture_number"] == pn]
        rr = rr[np.argsort(rr["lcu_index"])]
        assert len(rr) == nl, (name, pn, len(rr), nl)
        assert S.plane_checksum(S.gen_luma(kind, w, h, pn, seed)) == int(rr[0]["luma_crc"]), "input mismatch"
        assert all(rr["params"] == rr[0]["params"])
        if name in ROWS_KEPT:  # the tests run the whole picture and compare these LCUs only
            wl = (w + 63) // 64
            rr = rr[np.isin(rr["lcu_index"] // wl, ROWS_KEPT[name])]
        meta.append([pn, int(rr[0]["slice_type"]), int(rr[0]["enc_mode"])])
        params.append(rr[0]["params"])
        me_sad.append(rr["me_sad"])
        before.append(rr["before"])
        after.append(rr["after"])
    path = os.path.join(S.GOLDEN_DIR, "ois_%s.npz" % name)
    np.savez_compressed(path, clip=np.array([kind, str(w), str(h), str(n), str(seed)]), enc_args=np.array(args),
                        meta=np.array(meta, np.int64), params=np.array(params, S.OIS_PARAMS_DTYPE),
                        me_sad=np.stack(me_sad), before=np.stack(before), after=np.stack(after),
                        **({"rows_kept": np.array(ROWS_KEPT[name])} if name in ROWS_KEPT else {}))
    print("%-20s %d pictures x %d LCUs -> %s (%.0f KiB)  slice types %s" %
          (name, len(pics), nl, os.path.basename(path), os.path.getsize(path) / 1024, [m[1] for m in meta]))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
