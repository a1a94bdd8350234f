#!/usr/bin/env python3
"""Generate the encode-pass inter-prediction golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_INTER_DUMP set, so the --wrap interposer of
oracle/ref_harness_inter_dump.c records the padded reference pictures and a sample of the EncodePassInterPrediction calls
(motion vectors, direction, geometry -> the three predicted blocks).  Stored as tests/golden/inter_<name>.npz.
The 10-bit cases (names starting with "hbd_") go through oracle/ref_harness_inter16_dump.c (EncodePassInterPrediction16bit,
16-bit samples throughout) -> tests/golden/inter16_<name>.npz.
Needs /root/reference (this container only).  Usage: python tests/golden/make_inter_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

PIC = np.dtype([("magic", "<u4"), ("id", "<u4"), ("strideY", "<u4"), ("strideC", "<u4"), ("originX", "<u4"), ("originY", "<u4"),
                ("width", "<u4"), ("height", "<u4"), ("rowsY", "<u4"), ("rowsC", "<u4")])
UNIT = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("mv", "<i2", (2, 2)), ("pu_x", "<u2"), ("pu_y", "<u2"), ("pu_w", "u1"),
                 ("pu_h", "u1"), ("pred_dir", "u1"), ("pad", "u1"), ("ref_id", "<i4", 2), ("pred_y", "u1", 4096), ("pred_cb", "u1", 1024),
                 ("pred_cr", "u1", 1024)])

UNIT16 = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("mv", "<i2", (2, 2)), ("pu_x", "<u2"), ("pu_y", "<u2"), ("pu_w", "u1"),
                   ("pu_h", "u1"), ("pred_dir", "u1"), ("pad", "u1"), ("ref_id", "<i4", 2), ("pred_y", "<u2", 4096),
                   ("pred_cb", "<u2", 1024), ("pred_cr", "<u2", 1024)])

# name -> (clip kind, width, height, frames, seed, encoder args, sampling stride, units kept)
CASES = {
    "b_320x192_m7": ("motion", 320, 192, 9, 7, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"], 1, 250),
    "subpel_p_320x192_m3": ("subpel", 320, 192, 5, 3, ["-encMode", "3", "-pred-struct", "0", "-hierarchical-levels", "0"], 1, 300),
    "subpel_b_320x192_m6": ("subpel", 320, 192, 9, 3, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2"], 1, 300),
    "noise_b_320x256_m4": ("noise", 320, 256, 9, 11, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2"], 2, 300),
    # 10-bit (16-bit sample path)
    "hbd_subpel_b_256x128_m6": ("subpel", 256, 128, 9, 3, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-bit-depth", "10"], 1, 300),
    "hbd_subpel_p_256x128_m3": ("subpel", 256, 128, 5, 3, ["-encMode", "3", "-pred-struct", "0", "-hierarchical-levels", "0", "-bit-depth", "10"], 1, 250),
}


def write_subpel_clip(path, w, h, n, seed, depth=8):
    """smooth texture drifting by (5, 3) quarter samples per frame (plus a slow zoom): fractional motion vectors"""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 256, ((h + 64) // 4 + 2, (w + 64) // 4 + 2)).astype(np.float64)
    big = np.kron(big, np.ones((16, 16)))                      # 4x4 blocks at 4x oversampling
    k = np.ones(9) / 9
    for ax in (0, 1):                                          # blur so that sub-sample shifts matter
        big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
    with open(path, "wb") as f:
        for t in range(n):
            oy, ox = 32 + 3 * t, 32 + 5 * t                    # quarter-sample offsets into the 4x grid
            y = big[oy:oy + 4 * h:4, ox:ox + 4 * w:4]
            cb = 128 + 0.25 * (big[oy + 8:oy + 8 + 4 * h:8, ox:ox + 4 * w:8] - 128)
            cr = 128 - 0.25 * (big[oy:oy + 4 * h:8, ox + 8:ox + 8 + 4 * w:8] - 128)
            for pl in (y, cb, cr):
                if depth == 8:
                    f.write(np.clip(np.rint(pl), 0, 255).astype(np.uint8).tobytes())
                else:   # unpacked 10-bit: 16-bit little-endian samples
                    f.write(np.clip(np.rint(pl * 4), 0, 1023).astype("<u2").tobytes())


def parse(dump, hbd=False):
    raw = open(dump, "rb").read()
    pos, pics, units = 0, {}, []
    dt, bps, unit = (np.dtype("<u2"), 2, UNIT16) if hbd else (np.dtype(np.uint8), 1, UNIT)
    while pos < len(raw):
        magic = int(np.frombuffer(raw, "<u4", 1, pos)[0])
        if magic == (0x36495049 if hbd else 0x43495049):
            h = np.frombuffer(raw, PIC, 1, pos)[0]
            pos += PIC.itemsize
            ny, nc = int(h["rowsY"]) * int(h["strideY"]), int(h["rowsC"]) * int(h["strideC"])
            y = np.frombuffer(raw, dt, ny, pos).reshape(int(h["rowsY"]), int(h["strideY"])); pos += ny * bps
            cb = np.frombuffer(raw, dt, nc, pos).reshape(int(h["rowsC"]), int(h["strideC"])); pos += nc * bps
            cr = np.frombuffer(raw, dt, nc, pos).reshape(int(h["rowsC"]), int(h["strideC"])); pos += nc * bps
            pics[int(h["id"])] = (h, y.copy(), cb.copy(), cr.copy())
        else:
            assert magic == (0x364e5549 if hbd else 0x544e5549), hex(magic)
            units.append(np.frombuffer(raw, unit, 1, pos)[0])
            pos += unit.itemsize
    return pics, np.array(units)


def run_case(name):
    kind, w, h, n, seed, args, stride, keep = CASES[name]
    hbd = name.startswith("hbd_")
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "inter.dump")
        if kind == "subpel":
            write_subpel_clip(yuv, w, h, n, seed, 10 if hbd else 8)
        elif hbd:
            S.write_clip10(yuv, kind, w, h, n, seed)
        else:
            S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0",
               "-b", os.path.join(td, "out.265")] + args
        tag = "SVT_REF_INTER16" if hbd else "SVT_REF_INTER"
        subprocess.run(cmd, env=dict(os.environ, **{tag + "_DUMP": dump, tag + "_STRIDE": str(stride)}), check=True,
                       stdout=subprocess.DEVNULL)
        pics, units = parse(dump, hbd)
    assert len(units) and len(pics)
    # keep a spread of sizes / directions / fractional positions, and at most 4 reference pictures
    use = sorted(set(int(v) for v in units["ref_id"].reshape(-1) if v >= 0))[:3 if hbd else 4]
    ok = np.array([all(int(v) < 0 or int(v) in use for v in u["ref_id"]) for u in units])
    units = units[ok]
    frac = (units["mv"][:, 0, 0] & 3) + 4 * (units["mv"][:, 0, 1] & 3)
    order = np.lexsort((frac, units["pred_dir"], units["pu_w"]))
    sel = np.sort(np.unique(order[np.linspace(0, len(order) - 1, min(keep, len(order))).astype(int)]))
    units = units[sel]
    out = {k: units[k] for k in ("mv", "pu_x", "pu_y", "pu_w", "pu_h", "pred_dir", "ref_id")}
    out["pred_y"] = np.concatenate([u["pred_y"][: int(u["pu_w"]) * int(u["pu_h"])] for u in units])
    out["pred_cb"] = np.concatenate([u["pred_cb"][: int(u["pu_w"]) * int(u["pu_h"]) // 4] for u in units])
    out["pred_cr"] = np.concatenate([u["pred_cr"][: int(u["pu_w"]) * int(u["pu_h"]) // 4] for u in units])
    out["pic_ids"] = np.array(use)
    for i in use:
        h0, y, cb, cr = pics[i]
        out["pic%d_hdr" % i], out["pic%d_y" % i], out["pic%d_cb" % i], out["pic%d_cr" % i] = np.array([h0]), y, cb, cr
    path = os.path.join(S.GOLDEN_DIR, ("inter16_%s.npz" % name[4:]) if hbd else ("inter_%s.npz" % name))
    np.savez_compressed(path, **out)
    sizes, cnt = np.unique(units["pu_w"], return_counts=True)
    print("%-20s %d units (widths %s, directions %s, %d distinct luma fractions, |mv| max %d) over %d reference pictures -> %s (%.0f KiB)" %
          (name, len(units), dict(zip(sizes.tolist(), cnt.tolist())), np.unique(units["pred_dir"]).tolist(), len(np.unique(frac[sel])),
           int(np.abs(units["mv"]).max()), len(use), os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
