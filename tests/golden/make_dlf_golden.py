#!/usr/bin/env python3
"""Generate the picture-level deblocking + SAO-application golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref on seeded clips with SVT_REF_DLF_DUMP set, so the --wrap interposers of
oracle/ref_harness_dlf_dump.c record, per picture, the reconstruction before deblocking, the boundary-strength arrays,
the qp array and the reconstruction after the last LCU's deblocking drivers (SAO not applied yet); with SVT_REF_SAO_DUMP
oracle/ref_harness_sao_dump.c adds every LCU's final SAO parameters, and the encoder's own reconstruction output (-o)
gives the picture after SAO.
Stored as tests/golden/dlf_<name>.npz.  Needs /root/reference (this container only).
Usage: python tests/golden/make_dlf_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

HDR = np.dtype([("magic", "<u4"), ("header_size", "<u4"), ("picture_number", "<u8"), ("width", "<u4"), ("height", "<u4"),
                ("bytes_per_sample", "<u4"), ("slice_type", "<u4"), ("lcu_cols", "<u4"), ("lcu_rows", "<u4"),
                ("qp_stride", "<u4"), ("qp_size", "<u4"), ("tc_offset", "<i4"), ("beta_offset", "<i4"),
                ("cb_qp_offset", "<i4"), ("cr_qp_offset", "<i4")])

CUMAP = np.dtype([("mode", "u1"), ("dir", "u1"), ("size_log2", "u1"), ("pad", "u1"), ("mv", "<i2", (2, 2))])
SAO_HDR = np.dtype([("magic", "<u4"), ("header_size", "<u4"), ("picture_number", "<u8"), ("nlcu", "<u4"),
                    ("sao_flag", "<u4", 2), ("pad", "<u4")])
SAO_LCU = np.dtype([("merge_left", "u1"), ("merge_up", "u1"), ("edge_flags", "u1"), ("pad", "u1"), ("type", "<u4", 2),
                    ("offset", "<i4", (3, 4)), ("band", "<u4", 3)])

# name -> (clip kind, width, height, frames, seed, bit depth, encoder args, pictures kept)
CASES = {
    "p_416x240_m9": ("motion", 416, 240, 4, 7, 8, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0"], 3),
    "b_416x240_m7_q40": ("motion", 416, 240, 9, 7, 8, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "40"], 3),
    "noise_320x200_m6_q45": ("noise", 320, 200, 3, 11, 8, ["-encMode", "6", "-pred-struct", "1", "-hierarchical-levels", "0", "-q", "45"], 2),
    "p10_416x240_m9": ("motion", 416, 240, 3, 7, 10, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0", "-bit-depth", "10"], 2),
    "tiles_640x384_m9": ("motion", 640, 384, 3, 7, 8, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0", "-tile_row_cnt", "2", "-tile_col_cnt", "2"], 2),
}


def parse(dump):
    raw = open(dump, "rb").read()
    pos, recs = 0, []
    while pos < len(raw):
        h = np.frombuffer(raw, HDR, 1, pos)[0]
        assert h["magic"] == 0x20464c44 and h["header_size"] == HDR.itemsize
        pos += HDR.itemsize
        w, hh, bps = int(h["width"]), int(h["height"]), int(h["bytes_per_sample"])
        dt = np.uint8 if bps == 1 else np.dtype("<u2")
        planes = []
        for _ in range(2):
            for (pw, ph) in ((w, hh), (w // 2, hh // 2), (w // 2, hh // 2)):
                planes.append(np.frombuffer(raw, dt, pw * ph, pos).reshape(ph, pw).copy())
                pos += pw * ph * bps
        nlcu = int(h["lcu_cols"]) * int(h["lcu_rows"])
        bsv = np.frombuffer(raw, np.uint8, nlcu * 256, pos).reshape(nlcu, 256).copy(); pos += nlcu * 256
        bsh = np.frombuffer(raw, np.uint8, nlcu * 256, pos).reshape(nlcu, 256).copy(); pos += nlcu * 256
        qp = np.frombuffer(raw, np.uint8, int(h["qp_size"]), pos).copy(); pos += int(h["qp_size"])
        tag = np.frombuffer(raw, "<u4", 4, pos); pos += 16
        assert tag[0] == 0x31585342, hex(int(tag[0]))
        cumap = np.frombuffer(raw, CUMAP, int(tag[1]), pos).copy(); pos += CUMAP.itemsize * int(tag[1])
        cbf = np.frombuffer(raw, np.uint8, int(tag[2]), pos).copy(); pos += int(tag[2])
        refpoc = np.frombuffer(raw, "<u8", 2, pos).copy(); pos += 16
        edge = np.frombuffer(raw, np.uint8, int(tag[3]), pos).copy(); pos += int(tag[3])
        recs.append((h, planes, bsv, bsh, qp, cumap, cbf, refpoc, edge))
    return recs


def parse_sao(dump):
    """-> {picture_number: (sao_flag[2], per-LCU records)}"""
    out = {}
    if not os.path.exists(dump):
        return out
    raw = open(dump, "rb").read()
    pos = 0
    while pos < len(raw):
        h = np.frombuffer(raw, SAO_HDR, 1, pos)[0]
        assert h["magic"] == 0x204f4153 and h["header_size"] == SAO_HDR.itemsize
        pos += SAO_HDR.itemsize
        out[int(h["picture_number"])] = (h["sao_flag"].copy(), np.frombuffer(raw, SAO_LCU, int(h["nlcu"]), pos).copy())
        pos += SAO_LCU.itemsize * int(h["nlcu"])
    return out


def read_recon(path, w, h, bps, index):
    dt = np.uint8 if bps == 1 else np.dtype("<u2")
    frame = (w * h * 3 // 2) * bps
    raw = np.fromfile(path, dt, w * h * 3 // 2, offset=frame * index)
    y, cb, cr = raw[: w * h], raw[w * h: w * h * 5 // 4], raw[w * h * 5 // 4:]
    return [y.reshape(h, w), cb.reshape(h // 2, w // 2), cr.reshape(h // 2, w // 2)]


def run_case(name):
    kind, w, h, n, seed, depth, args, keep = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "dlf.dump")
        (S.write_clip10 if depth == 10 else S.write_clip)(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-q", "32", "-asm", "0", "-sao", "1",
               "-b", os.path.join(td, "out.265"), "-o", os.path.join(td, "recon.yuv")] + args  # -o: every picture deblocked
        sdump = os.path.join(td, "sao.dump")
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_DLF_DUMP=dump, SVT_REF_SAO_DUMP=sdump), check=True,
                       stdout=subprocess.DEVNULL)
        recs = parse(dump)
        sao = parse_sao(sdump)
        finals = {int(r[0]["picture_number"]): read_recon(os.path.join(td, "recon.yuv"), w, h, int(r[0]["bytes_per_sample"]),
                                                          int(r[0]["picture_number"])) for r in recs}
    assert len(recs) >= 2, (len(recs), n)   # pictures that may mismatch the decoder (non-reference, top layer) skip the filter
    recs.sort(key=lambda r: int(r[0]["picture_number"]))
    # keep the pictures with the most filtered samples, but always one inter picture
    changed = [int(sum((a != b).sum() for a, b in zip(r[1][:3], r[1][3:]))) for r in recs]
    order = sorted(range(len(recs)), key=lambda i: -changed[i])[:keep]
    out, sao_changed = {}, {}
    for k, i in enumerate(sorted(order)):
        h0, planes, bsv, bsh, qp, cumap, cbf, refpoc, edge = recs[i]
        out["cumap%d" % k], out["cbf%d" % k], out["refpoc%d" % k], out["lcu_edge%d" % k] = cumap, cbf, refpoc, edge
        out["hdr%d" % k] = np.array([h0])
        for nm, a in zip(("pre_y", "pre_cb", "pre_cr", "post_y", "post_cb", "post_cr"), planes):
            out["%s%d" % (nm, k)] = a
        out["bsv%d" % k], out["bsh%d" % k], out["qp%d" % k] = bsv, bsh, qp
        pn = int(h0["picture_number"])
        flags, lcus = sao.get(pn, (np.zeros(2, np.uint32), np.zeros(len(bsv), SAO_LCU)))
        out["sao_flag%d" % k], out["sao_lcu%d" % k] = flags, lcus
        for nm, a, b in zip(("final_y", "final_cb", "final_cr"), finals[pn], planes[3:]):
            out["%s%d" % (nm, k)] = a
            sao_changed[k] = sao_changed.get(k, 0) + int((a != b).sum())
    out["count"] = np.array(len(order))
    path = os.path.join(S.GOLDEN_DIR, "dlf_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-24s %d pictures kept of %d -> %s (%.0f KiB); changed samples dlf %s sao %s, slice types %s, bS values %s, qp %s" %
          (name, len(order), n, os.path.basename(path), os.path.getsize(path) / 1024, [changed[i] for i in sorted(order)],
           [sao_changed[k] for k in sorted(sao_changed)],
           [int(recs[i][0]["slice_type"]) for i in sorted(order)],
           np.unique(np.concatenate([recs[i][2].reshape(-1) for i in order])).tolist(),
           np.unique(np.concatenate([recs[i][4] for i in order])).tolist()))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
