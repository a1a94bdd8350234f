"""CPU-only: the mode-decision checker (oracle/svt_oracle_md.c = the scalar decisions of svt-hevc_amd/csrc/md_logic.h, the text the HIP kernel
executes, composed with the pinned oracle leaves) against whole pictures of ModeDecisionLcu calls recorded inside the reference encoder
(tests/golden/md_*.npz, oracle/ref_harness_md_dump.c): for every leaf the reference tested - split flag, prediction mode, intra luma mode,
luma cbf and the cost the inter-depth decisions compared."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "md_*.npz")))


def oracle_md_picture(lib, g, k):
    pic = np.ascontiguousarray(g["pic"][k:k + 1])
    lcus = np.ascontiguousarray(g["lcu"][k])
    cost = np.ascontiguousarray(g["cost"][k])
    src = np.ascontiguousarray(g["src_y"][k])
    ois = np.ascontiguousarray(g["ois"][k])
    out = np.zeros(len(lcus), S.MD_LCU_OUT_DTYPE)
    rec = np.zeros_like(src)
    if "inter" in g.files:
        X, me, tmvp, refs, keep = inter_inputs(g, k)
        cb, cr = np.ascontiguousarray(g["src_cb"][k]), np.ascontiguousarray(g["src_cr"][k])
        kinds = np.zeros((len(lcus), 85), np.uint8)
        lib.svt_oracle_md_picture_inter.restype = C.c_int
        lib.svt_oracle_md_picture_inter.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 8
        rc = lib.svt_oracle_md_picture_inter(pic.ctypes.data, X.ctypes.data, lcus.ctypes.data, cost.ctypes.data, src.ctypes.data, src.shape[1],
                                             cb.ctypes.data, cr.ctypes.data, cb.shape[1], ois.ctypes.data, me.ctypes.data,
                                             tmvp.ctypes.data if tmvp is not None else None, C.addressof(refs[0]), C.addressof(refs[1]), out.ctypes.data,
                                             rec.ctypes.data, kinds.ctypes.data)
        assert rc == 0, rc
        oracle_md_picture.kinds = kinds
        return out, rec
    lib.svt_oracle_md_picture.restype = C.c_int
    lib.svt_oracle_md_picture.argtypes = [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 3
    rc = lib.svt_oracle_md_picture(pic.ctypes.data, lcus.ctypes.data, cost.ctypes.data, src.ctypes.data, src.shape[1], ois.ctypes.data,
                                   out.ctypes.data, rec.ctypes.data)
    assert rc == 0, rc
    return out, rec


def inter_inputs(g, k):
    """the inter inputs of picture k of a P / B fixture: SvtAmdMdInter, the ME records in the contract's layout, the co-located motion field and
    the two reference pictures (host planes) as SvtAmdRefPicture"""
    X = np.ascontiguousarray(g["inter"][k:k + 1])
    nl = g["lcu"].shape[1]
    me = np.zeros(nl, S.ME_LCU_DTYPE)
    me["pu"] = g["me"][k]
    tmvp = np.ascontiguousarray(g["tmvp"][k]) if g["tmvp_present"][k] else None
    sy, sc, ox, oy, rw, rh, _ = (int(v) for v in g["ref_geom"][k])
    keep, refs = [], []
    for l in range(2):
        planes = [np.ascontiguousarray(g["ref%d_%s" % (l, nm)][k]) for nm in ("y", "cb", "cr")]
        keep.append(planes)
        refs.append(S.RefPicture(planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, sy, sc, ox, oy, rw, rh))
    return X, me, tmvp, refs, keep


def compare_md(got, want, what):
    """every leaf the reference tested: same test set, and for tested leaves the same decisions and costs"""
    errs = []
    for i in range(len(want)):
        g, w = got[i], want[i]
        if not np.array_equal(g["tested"], w["tested"]):
            errs.append((i, "tested", np.nonzero(g["tested"] != w["tested"])[0][:6].tolist()))
            continue
        t = w["tested"] == 1
        inter = t & (w["pred_mode"] == 1)
        merge = inter & (w["merge_flag"] == 1)
        for f, m in (("split", t), ("pred_mode", t), ("intra_luma_mode", t), ("ycbf", t), ("cost", t), ("inter_dir", t), ("merge_flag", t),
                     ("merge_index", merge), ("mv", inter), ("merge_cost", merge), ("skip_cost", merge)):
            if not np.array_equal(g[f][m], w[f][m]):
                ne = g[f] != w[f]
                bad = np.nonzero((ne.reshape(85, -1).any(axis=1)) & m)[0]
                errs.append((i, f, bad[:6].tolist(), g[f][bad[:3]].tolist(), w[f][bad[:3]].tolist()))
    assert not errs, "%s: %d LCU fields differ, first: %s" % (what, len(errs), errs[:4])


def compare_kinds(got, want, what):
    known = want != 0xFE
    assert known.any(), what
    bad = np.argwhere((got != want) & known)
    assert len(bad) == 0, "%s: %d inter units with another merge / skip decision, first (lcu, leaf): %s got %s want %s" % (
        what, len(bad), bad[:5].tolist(), [int(got[i, j]) for i, j in bad[:5]], [int(want[i, j]) for i, j in bad[:5]])


@pytest.mark.parametrize("name", CASES)
def test_oracle_md_matches_recorded_mode_decisions(name):
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    lib = S.load_oracle()
    assert len(g["picture_number"]) >= 1
    for k in range(len(g["picture_number"])):
        out, _ = oracle_md_picture(lib, g, k)
        compare_md(out, g["out"][k], "%s picture %d" % (name, int(g["picture_number"][k])))
        if "inter" in g.files:   # what EncodePass does with the inter units of the final trees: AMVP / merge / skip (its chroma-completed costs)
            compare_kinds(oracle_md_picture.kinds, g["ep_kind"][k], "%s picture %d" % (name, int(g["picture_number"][k])))


# ---- the EncDec input contract the decisions amount to (what the device builds behind its mode decision; test-side restatement) ----
def md_stats(leaf):
    """GetCodedUnitStats: (depth, size, x, y) of a leaf of the depth-first scan"""
    if leaf == 0:
        return 0, 64, 0, 0
    r = leaf - 1
    q, r32 = divmod(r, 21)
    x, y = (q & 1) * 32, (q >> 1) * 32
    if r32 == 0:
        return 1, 32, x, y
    s, r16 = divmod(r32 - 1, 5)
    x, y = x + (s & 1) * 16, y + (s >> 1) * 16
    if r16 == 0:
        return 2, 16, x, y
    e = r16 - 1
    return 3, 8, x + (e & 1) * 8, y + (e >> 1) * 8


def final_tree(out_lcu, lw=64, lh=64):
    """the leaves with split == 0 in Z order, as EncodePass walks them (EbCodingLoop.c:3180, :4586-4592)"""
    units, it = [], 0
    while it < 85:
        if out_lcu["split"][it]:
            it += 1
            continue
        d, s, x, y = md_stats(it)
        if x < lw and y < lh:
            units.append((it, x, y, s))
        it += (85, 21, 5, 1)[d]
    return units


def works_from_md(pic, lcus, out, src):
    from test_gpu_encodepass import z_available
    w, h = int(pic["width"]), int(pic["height"])
    wl = (w + 63) // 64
    works = np.zeros(len(lcus), S.LCU_WORK_DTYPE)
    for k in range(len(lcus)):
        lx, ly = 64 * (k % wl), 64 * (k // wl)
        lw, lh = min(64, w - lx), min(64, h - ly)
        wk = works[k]
        wk["lcu_x"], wk["lcu_y"], wk["slice_type"], wk["temporal_layer"] = lx, ly, pic["slice_type"], pic["temporal_layer"]
        wk["constrained_intra"], wk["strong_smoothing"] = pic["constrained_intra"], pic["strong_smoothing"]
        wk["tile_left"], wk["tile_top"], wk["tile_right"] = lcus[k]["tile_left"], lcus[k]["tile_top"], lcus[k]["tile_right"]
        wk["full_lambda"] = pic["full_lambda"]
        wk["luma_cbf_bits"] = [pic["rates"]["lumaCbfBits"][i] for i in (0, 1, 5, 6)]
        units = final_tree(out[k], lw, lh)
        wk["num_cus"] = len(units)
        for i, (leaf, x, y, s) in enumerate(units):
            cu = wk["cu"][i]
            cu["x"], cu["y"], cu["size"], cu["pred_mode"], cu["intra_luma_mode"] = x, y, s, out[k]["pred_mode"][leaf], out[k]["intra_luma_mode"][leaf]
            cu["bottom_left_ok"], cu["top_right_ok"] = z_available(x, y, s)
            cu["qp"], cu["chroma_qp"], cu["leaf_index"] = pic["qp"], pic["chroma_qp"], leaf
        sy = np.zeros((64, 64), np.uint8)
        sy[:lh, :lw] = src[0][ly:ly + lh, lx:lx + lw]
        wk["src_y"] = sy.reshape(-1)
        for p, nm in ((1, "src_cb"), (2, "src_cr")):
            sc = np.zeros((32, 32), np.uint8)
            sc[:lh // 2, :lw // 2] = src[p][ly // 2:(ly + lh) // 2, lx // 2:(lx + lw) // 2]
            wk[nm] = sc.reshape(-1)
    return works


def test_recorded_decisions_are_the_trees_the_encode_pass_fixture_holds():
    """cross-check of two independent recordings of the same encode: the final trees of md_i_motion_416x240_m9 (ModeDecisionLcu records) are the
    coding-unit lists of encodepass_i_motion_416x240_m9 (EncodePass records)"""
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_i_motion_416x240_m9.npz"))
    e = np.load(os.path.join(S.GOLDEN_DIR, "encodepass_i_motion_416x240_m9.npz"))
    assert list(g["enc_args"]) == list(e["enc_args"])
    nl = g["lcu"].shape[1]
    for k in range(len(g["picture_number"])):
        works = works_from_md(g["pic"][k], g["lcu"][k], g["out"][k], (g["src_y"][k], g["src_cb"][k], g["src_cr"][k]))
        ew = e["work"][k * nl:(k + 1) * nl]
        for i in range(nl):
            n = int(ew[i]["num_cus"])
            assert int(works[i]["num_cus"]) == n, (k, i)
            for f in ("x", "y", "size", "pred_mode", "intra_luma_mode", "bottom_left_ok", "top_right_ok", "qp", "chroma_qp", "leaf_index"):
                assert np.array_equal(works[i]["cu"][f][:n], ew[i]["cu"][f][:n]), (k, i, f)
            for f in ("src_y", "src_cb", "src_cr", "lcu_x", "lcu_y", "tile_left", "tile_top", "tile_right", "slice_type", "strong_smoothing", "constrained_intra"):
                assert np.array_equal(works[i][f], ew[i][f]), (k, i, f)


MV_BITS_DIGEST = os.path.join(S.GOLDEN_DIR, "mv_bit_table.sha256")


def test_mv_bit_table():
    """md_mv_bits (md_logic.h: a 3 x 3 core + 2 bits per doubling, selects instead of a table in memory) against the reference's
    mvBitTable[500][500] (Codec/EbModeDecisionConfiguration.h:108): entry by entry when the reference is built here
    (oracle/_ref, svt_ref_mv_bits), and by the digest of the whole table committed from such a run (tests/golden/mv_bit_table.sha256)."""
    import hashlib
    lib = S.load_oracle()
    lib.svt_oracle_md_mv_bits.restype = C.c_uint32
    lib.svt_oracle_md_mv_bits.argtypes = [C.c_int, C.c_int]
    ours = np.array([[lib.svt_oracle_md_mv_bits(x, y) for y in range(500)] for x in range(500)], np.uint32)
    ref = S.load_ref()
    if ref is not None and hasattr(ref, "svt_ref_mv_bits"):
        ref.svt_ref_mv_bits.restype = C.c_uint32
        ref.svt_ref_mv_bits.argtypes = [C.c_int, C.c_int]
        theirs = np.array([[ref.svt_ref_mv_bits(x, y) for y in range(500)] for x in range(500)], np.uint32)
        bad = np.argwhere(ours != theirs)
        assert len(bad) == 0, "mvBitTable differs at %s: %d vs %d" % (bad[0], ours[tuple(bad[0])], theirs[tuple(bad[0])])
        digest = hashlib.sha256(theirs.astype("<u4").tobytes()).hexdigest()
        if not os.path.exists(MV_BITS_DIGEST):
            open(MV_BITS_DIGEST, "w").write(digest + "\n")
        assert open(MV_BITS_DIGEST).read().strip() == digest
    assert os.path.exists(MV_BITS_DIGEST), "no reference here and no committed digest"
    assert hashlib.sha256(ours.astype("<u4").tobytes()).hexdigest() == open(MV_BITS_DIGEST).read().strip()
