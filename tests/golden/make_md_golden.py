#!/usr/bin/env python3
"""Generate the mode-decision golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref with SVT_REF_MD_DUMP set: the --wrap interposer oracle/ref_harness_md_dump.c records, for every picture
whose LCUs go through ModeDecisionLcu, the picture-level controls + rate tables + source picture + open-loop intra search results, and per
LCU the controls (MDC leaf list, detector flags) and what the reference's mode decision left (split flags, modes, luma cbf, costs).
Stored as tests/golden/md_<name>.npz.  Needs /root/reference.  Usage: make_md_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

CASES = {
    # name -> (clip, w, h, frames, seed, args, keep): keep = which recorded pictures go into the fixture
    # all-intra, encMode 9: 35-mode injection (intraInjectionMethod 1), MPM search with one candidate; partial right column and bottom row
    "i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "30"], None),
    # BASELINE configs[0] class: encMode 10 (OIS candidate lists, intraInjectionMethod 0)
    "i_motion_1920x1080_m10": ("motion", 1920, 1080, 1, 7, ["-encMode", "10", "-intra-period", "0", "-q", "32"], None),
    # noise at a low qp: every unit carries coefficients, small units win
    "i_noise_320x256_m8_q22": ("noise", 320, 256, 1, 11, ["-encMode", "8", "-intra-period", "0", "-q", "22"], None),
    # flat + high qp: 32x32 units without coefficients, anti-contouring classes
    "i_flat_320x192_m9_q44": ("flat", 320, 192, 1, 5, ["-encMode", "9", "-intra-period", "0", "-q", "44"], None),
    # 2x2 tiles: neighbour arrays per tile
    "i_tiles_motion_640x384_m9": ("motion", 640, 384, 1, 7, ["-encMode", "9", "-intra-period", "0", "-q", "33", "-tile_col_cnt", "2", "-tile_row_cnt", "2"], None),
}


def parse_dump(raw):
    pics, lcus, off = {}, [], 0
    while off < len(raw):
        magic, size = np.frombuffer(raw, "<u4", 2, off)
        if magic == S.MD_PIC_MAGIC:
            h = np.frombuffer(raw, S.MD_PIC_RECORD_DTYPE, 1, off)[0]
            w, hh, nl = int(h["pic"]["width"]), int(h["pic"]["height"]), int(h["nlcu"])
            o = off + S.MD_PIC_RECORD_DTYPE.itemsize
            y = np.frombuffer(raw, np.uint8, w * hh, o).reshape(hh, w)
            cb = np.frombuffer(raw, np.uint8, w * hh // 4, o + w * hh).reshape(hh // 2, w // 2)
            cr = np.frombuffer(raw, np.uint8, w * hh // 4, o + w * hh * 5 // 4).reshape(hh // 2, w // 2)
            ois = np.frombuffer(raw, S.OIS_LCU_DTYPE, nl, o + w * hh * 3 // 2)
            assert S.MD_PIC_RECORD_DTYPE.itemsize + w * hh * 3 // 2 + nl * S.OIS_LCU_DTYPE.itemsize == size
            pics[int(h["picture_number"])] = (h, y, cb, cr, ois)
        elif magic == S.MD_LCU_MAGIC:
            assert size == S.MD_LCU_RECORD_DTYPE.itemsize, (size, S.MD_LCU_RECORD_DTYPE.itemsize)
            lcus.append(np.frombuffer(raw, S.MD_LCU_RECORD_DTYPE, 1, off)[0])
        else:
            raise AssertionError("bad record magic %x at %d" % (magic, off))
        off += int(size)
    return pics, np.array(lcus, dtype=S.MD_LCU_RECORD_DTYPE)


def run_case(name):
    kind, w, h, n, seed, args, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "md.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_MD_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        pics, lcus = parse_dump(open(dump, "rb").read())
    nl = S.lcu_count(w, h)
    numbers = sorted(pics) if keep is None else [p for p in sorted(pics) if p in keep]
    # only pictures every LCU of which went through ModeDecisionLcu (PICT_LCU_SWITCH pictures mix it with the BDP path)
    numbers = [p for p in numbers if (lcus["picture_number"] == p).sum() == nl]
    assert numbers, "no picture with all %d LCUs recorded" % nl
    out = {"clip": np.array([kind, str(w), str(h), str(n), str(seed)]), "enc_args": np.array(args), "picture_number": np.array(numbers, np.uint64)}
    recs = []
    for p in numbers:
        r = lcus[lcus["picture_number"] == p]
        r = r[np.argsort(r["lcu_index"])]
        assert np.array_equal(r["lcu_index"], np.arange(nl))
        recs.append(r)
    out["pic"] = np.stack([pics[p][0]["pic"] for p in numbers])
    out["cost"] = np.stack([pics[p][0]["cost"] for p in numbers])
    out["src_y"] = np.stack([pics[p][1] for p in numbers])
    out["src_cb"] = np.stack([pics[p][2] for p in numbers])
    out["src_cr"] = np.stack([pics[p][3] for p in numbers])
    out["ois"] = np.stack([pics[p][4] for p in numbers])
    out["lcu"] = np.stack([r["lcu"] for r in recs])
    out["out"] = np.stack([r["out"] for r in recs])
    path = os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name)
    np.savez_compressed(path, **out)
    o = out["out"]
    final = (o["split"] == 0) & (o["tested"] == 1)
    print("%-28s pictures %s, %d LCUs each -> %s (%.0f KiB); slice types %s; tested leaves %d, unsplit tested leaves %d" %
          (name, numbers, nl, os.path.basename(path), os.path.getsize(path) / 1024, out["pic"]["slice_type"].tolist(), int(o["tested"].sum()),
           int(final.sum())))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
