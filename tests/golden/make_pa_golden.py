#!/usr/bin/env python3
"""Generate the picture-analysis fixtures from the REFERENCE itself: oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_PA_DUMP set
(oracle/ref_harness_me_dump.c records, at every MotionEstimateLcu call, what GatheringPictureStatistics left for that LCU - variance[85], yMean[85] -
and with LCU 0 the picture's luma histograms / region averages).  -> tests/golden/pa_<name>.npz.  Needs /root/reference (this container only).
Usage: python tests/golden/make_pa_golden.py [name ...]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

CASES = {
    # partial right column and bottom row (416 = 6.5, 240 = 3.75 LCUs): the statistics of partial LCUs read the padding
    "motion_416x240_m7": ("motion", 416, 240, 5, 7, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"]),
    "noise_320x256_m5": ("noise", 320, 256, 3, 11, ["-encMode", "5", "-pred-struct", "0"]),
    "objects_1280x720_m8": ("objects", 1280, 720, 3, 3, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "2"]),
}


def run_case(name):
    kind, w, h, n, seed, args = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "pa.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0", "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_PA_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        recs = np.fromfile(dump, dtype=S.PA_DUMP_DTYPE)
    assert len(recs) and (recs["magic"] == 0x50414453).all()
    nl = S.lcu_count(w, h)
    pics = sorted(set(int(p) for p in recs["picture_number"]))
    var, mean, hist, ravg, avg = [], [], [], [], []
    for p in pics:
        r = recs[recs["picture_number"] == p]
        # two lists = two calls per LCU with the same statistics: keep the first of each LCU
        _, first = np.unique(r["lcu_index"], return_index=True)
        r = r[first]
        assert np.array_equal(r["lcu_index"], np.arange(nl))
        var.append(r["variance"]), mean.append(r["y_mean"])
        head = r[0]
        assert head["kind"] == 1 and head["regions_w"] == 4 and head["regions_h"] == 4
        hist.append(head["histogram"]), ravg.append(head["region_average"]), avg.append(head["average_intensity"])
    out = dict(clip=np.array([kind, str(w), str(h), str(n), str(seed)]), enc_args=np.array(args), picture_number=np.array(pics, np.uint64),
               variance=np.stack(var), y_mean=np.stack(mean), histogram=np.stack(hist), region_average=np.stack(ravg), average_intensity=np.array(avg, np.uint8))
    path = os.path.join(S.GOLDEN_DIR, "pa_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-24s pictures %s, %d LCUs each -> %s (%d KiB)" % (name, pics, nl, os.path.basename(path), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
