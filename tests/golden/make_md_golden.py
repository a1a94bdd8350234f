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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
    # P / B pictures: the non-reference B pictures of random-access encodes (open-loop intra, luma-only candidates, partial frequency N2)
    "b_motion_416x240_m8": ("motion", 416, 240, 9, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "30"], "nonref"),
    "b_noise_320x256_m9_q24": ("noise", 320, 256, 9, 11, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "24"], "nonref"),
    "b_tiles_motion_640x384_m8": ("motion", 640, 384, 5, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "34", "-tile_col_cnt", "2",
                                                               "-tile_row_cnt", "2"], "nonref"),
    "b_objects_416x240_m8": ("objects", 416, 240, 9, 3, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "28"], "nonref"),
    "b_objects_640x360_m9_q36": ("objects", 640, 360, 5, 5, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "36"], "nonref"),
    "p_objects_320x192_m8_ld": ("objects", 320, 192, 6, 4, ["-encMode", "8", "-pred-struct", "1", "-hierarchical-levels", "2", "-q", "26"], "nonref"),
    "p_motion_416x240_m8_ld": ("motion", 416, 240, 6, 9, ["-encMode", "8", "-pred-struct", "0", "-hierarchical-levels", "2", "-q", "32"], "nonref"),
    # reference P / B pictures with open-loop intra candidates: chroma level 4 = per-LCU switch between CHROMA_MODE_FULL (chroma prediction + SAD in the fast loop, chroma
    # full loop and chroma terms in the full costs of every candidate) and CHROMA_MODE_BEST (EbEncDecProcess.c:2066-2113, EbModeDecisionProcess.c:395-447)
    "bref_motion_416x240_m8": ("motion", 416, 240, 9, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "30"], "ref"),
    "bref_objects_640x360_m9_q36": ("objects", 640, 360, 5, 5, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "36"], "ref"),
    "bref_noise_320x256_m9_q24": ("noise", 320, 256, 9, 11, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "24"], "ref"),
    "pref_motion_416x240_m8_ld": ("motion", 416, 240, 6, 9, ["-encMode", "8", "-pred-struct", "0", "-hierarchical-levels", "2", "-q", "32"], "ref"),
    "bref_tiles_motion_640x384_m8": ("motion", 640, 384, 9, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "34", "-tile_col_cnt", "2",
                                                                  "-tile_row_cnt", "2"], "ref"),
    "i_tiles_motion_640x384_m9": ("motion", 640, 384, 1, 7, ["-encMode", "9", "-intra-period", "0", "-q", "33", "-tile_col_cnt", "2", "-tile_row_cnt", "2"], None),
    # detector outcomes no synthetic clip produces, set on every 3rd / 2nd LCU by the harness before the recorded call (SVT_REF_MD_FORCE, oracle/ref_harness_md_dump.c):
    # LCU_COMPLEXITY_STATUS_2 LCUs (the complexity branch of the intra candidate injection) and CMPLX_NOISE LCUs (every LCU of the reference B pictures; the noise-class rule of the fast loop's chroma distortion
    # for 64x64 candidates that do not move: the zero-vector merge candidates of a static clip), in I, non-reference B and CHROMA_MODE_FULL reference B pictures
    "i_forced_motion_416x240_m9": ("motion", 416, 240, 1, 7, ["-encMode", "9", "-intra-period", "0", "-q", "32"], None, "complex2:3,noise:2"),
    "b_forced_static_416x240_m8": ("static", 416, 240, 9, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "30"], "nonref", "complex2:3,noise:2"),
    "bref_forced_static_416x240_m8": ("static", 416, 240, 9, 7, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "30"], "ref", "complex2:3,noise:1"),
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
            o += w * hh * 3 // 2 + nl * S.OIS_LCU_DTYPE.itemsize
            inter = None
            if h["has_inter"]:
                me = np.frombuffer(raw, S.ME_LCU_DTYPE, nl, o)
                o += nl * S.ME_LCU_DTYPE.itemsize
                tmvp = None
                if h["tmvp_present"]:
                    tmvp = np.frombuffer(raw, S.MD_TMVP_LCU_DTYPE, nl, o)
                    o += nl * S.MD_TMVP_LCU_DTYPE.itemsize
                sy, sc, oy, rh = int(h["ref_stride_y"]), int(h["ref_stride_c"]), int(h["ref_origin_y"]), int(h["ref_height"])
                refs = []
                for _ in range(int(h["nref"])):
                    ny, nc = sy * (rh + 2 * oy), sc * (rh // 2 + oy)
                    refs.append((np.frombuffer(raw, np.uint8, ny, o), np.frombuffer(raw, np.uint8, nc, o + ny), np.frombuffer(raw, np.uint8, nc, o + ny + nc)))
                    o += ny + 2 * nc
                inter = (me, tmvp, refs)
            assert o - off == size, (o - off, size)
            pics[int(h["picture_number"])] = (h, y, cb, cr, ois, inter)
        elif magic == S.MD_LCU_MAGIC:
            assert size == S.MD_LCU_RECORD_DTYPE.itemsize, (size, S.MD_LCU_RECORD_DTYPE.itemsize)
            lcus.append(np.frombuffer(raw, S.MD_LCU_RECORD_DTYPE, 1, off)[0])
        else:
            raise AssertionError("bad record magic %x at %d" % (magic, off))
        off += int(size)
    return pics, np.array(lcus, dtype=S.MD_LCU_RECORD_DTYPE)


def run_case(name):
    kind, w, h, n, seed, args, keep = CASES[name][:7]
    force = CASES[name][7] if len(CASES[name]) > 7 else None
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "md.dump")
        S.write_clip(yuv, kind, w, h, n, seed)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-b", os.path.join(td, "out.265")] + args
        epdump = os.path.join(td, "ep.dump")
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_MD_DUMP=dump, SVT_REF_ENCODEPASS_DUMP=epdump, **({"SVT_REF_MD_FORCE": force} if force else {})), check=True, stdout=subprocess.DEVNULL)
        pics, lcus = parse_dump(open(dump, "rb").read())
        import make_encodepass_golden as EPG
        ep_recs, _, _ = EPG.parse_dump(open(epdump, "rb").read(), S.EP_RECORD_DTYPE)
    nl = S.lcu_count(w, h)
    if os.environ.get("MD_GOLDEN_LIST"):  # survey: what the reference derived for every recorded picture
        for p in sorted(pics):
            h = pics[p][0]
            pc = h["pic"]
            r = lcus[lcus["picture_number"] == p]
            print("  poc %d: slice %d tl %d ref %d depth_mode %d open_loop %d chroma %d pf %d nfl %d nmm %d cabac_upd %d i4x4 %d mpm %d limit_intra %d cu8x8 %d "
                  "subpel %d tmvp %d lcus via MD %d/%d modes %s chroma modes %s" %
                  (p, pc["slice_type"], pc["temporal_layer"], pc["is_reference"], pc["depth_mode"], pc["intra_md_open_loop"], pc["chroma_level"], pc["pf_md_level"],
                   pc["nfl_level_md"], pc["nmm_level_md"], pc["coeff_cabac_update"], pc["intra4x4_level"], pc["mpm_search"], pc["limit_intra"], pc["cu8x8_mode"],
                   h["inter"]["use_subpel"], h["inter"]["tmvp_enable"], len(r), nl, sorted(set(r["lcu"]["lcu_md_mode"].tolist())),
                   sorted(set(r["lcu"]["chroma_encode_mode"].tolist()))))
    if keep == "nonref":
        keep = [p for p in sorted(pics) if pics[p][0]["pic"]["slice_type"] != 2 and not pics[p][0]["pic"]["is_reference"]]
    if keep == "ref":
        keep = [p for p in sorted(pics) if pics[p][0]["pic"]["slice_type"] != 2 and pics[p][0]["pic"]["is_reference"] and pics[p][0]["pic"]["intra_md_open_loop"]]
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
    # what EncodePass did with the inter units of the final trees (EncodePass records of the same run): per leaf SVT_AMD_EP_INTER_*, 0xFF = not an
    # inter unit of the final tree, 0xFE = the LCU has no EncodePass record
    epk = np.full((len(numbers), nl, 85), 0xFE, np.uint8)
    for k, p in enumerate(numbers):
        for r in ep_recs[ep_recs["picture_number"] == p]:
            row = epk[k, int(r["lcu_index"])]
            row[:] = 0xFF
            n = int(r["work"]["num_cus"])
            cu = r["work"]["cu"][:n]
            inter = cu["pred_mode"] == 1
            row[cu["leaf_index"][inter]] = cu["inter_kind"][inter]
    out["ep_kind"] = epk
    if any(pics[p][5] is not None for p in numbers):  # P / B pictures: the inter inputs (every kept picture must be one)
        assert all(pics[p][5] is not None for p in numbers)
        out["inter"] = np.stack([pics[p][0]["inter"] for p in numbers])
        out["me"] = np.stack([pics[p][5][0] for p in numbers])["pu"]  # only the candidate records matter to the mode decision
        out["tmvp_present"] = np.array([pics[p][5][1] is not None for p in numbers])
        out["tmvp"] = np.stack([pics[p][5][1] if pics[p][5][1] is not None else np.zeros(nl, S.MD_TMVP_LCU_DTYPE) for p in numbers])
        out["ref_geom"] = np.stack([np.array([pics[p][0][k] for k in ("ref_stride_y", "ref_stride_c", "ref_origin_x", "ref_origin_y", "ref_width",
                                                                       "ref_height", "nref")], np.uint32) for p in numbers])
        for l in range(2):
            for k, nm in enumerate(("y", "cb", "cr")):
                out["ref%d_%s" % (l, nm)] = np.stack([pics[p][5][2][min(l, len(pics[p][5][2]) - 1)][k] for p in numbers])
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
