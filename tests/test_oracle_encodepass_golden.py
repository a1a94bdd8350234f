"""CPU-only: pins oracle/svt_oracle_encodepass.c:svt_oracle_encode_lcu (the coding-unit loop of EncodePass for LCUs of intra 2Nx2N
units, neighbours read from the un-deblocked picture + a mode map) against records of real EncodePass calls of the reference
(tests/golden/encodepass_*.npz, made by tests/golden/make_encodepass_golden.py with the loop filters off)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

ALL = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "encodepass_*.npz")))
INTER_CASES = [c for c in ALL if c.split("_")[0] in ("p", "b", "p10", "b10")]   # P / B pictures: inter units, reference pictures, rate tables
SAO_CASES = [c for c in ALL if c.startswith("sao_")]        # deblocking and SAO on: decision records of every LCU + the encoder's finished output
CASES = [c for c in ALL if not c.startswith(("dlf_", "sao_")) and c not in INTER_CASES]  # all-intra, loop filters off: per-LCU reconstruction comparable
DLF_CASES = [c for c in ALL if c.startswith("dlf_")]        # deblocking on, SAO off: the encoder's output picture is the reference


def load_case(name):
    z = np.load(os.path.join(S.GOLDEN_DIR, "encodepass_%s.npz" % name))
    g = {k: z[k] for k in z.files}
    # the records in the CURRENT field names of the contract (fixtures written before a pad byte got a name have the same bytes)
    wide = g["work"].dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize
    g["work"] = np.ascontiguousarray(g["work"]).view(S.LCU_WORK16_DTYPE if wide else S.LCU_WORK_DTYPE)
    w, h = int(g["clip"][1]), int(g["clip"][2])
    return g, w, h


def is16(g):
    """10-bit encodes (EncodePass with is16bit) are recorded with 16-bit source / reconstruction samples"""
    return g["work"].dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize


def compare_lcu(work, want, got, w, h, tag, rec=True):
    """cbf / DC-only / counts of the units, the quantised coefficients of every unit area and the LCU's reconstruction inside the picture"""
    n = int(work["num_cus"])
    ne = 5 if int(work["cu"][0]["size"]) == 64 else n     # a 64x64 unit: entry 0 = the OR of its four transform units' flags, 1..4 = the units
    for f in ("cbf", "only_dc", "nz"):
        a, b = got["cu"][f][:ne], want["cu"][f][:ne]
        if ne == 5 and f != "cbf":
            a, b = a[1:], b[1:]                           # transformUnitArray[0] of a 64x64 unit carries flags only
        if f != "cbf":                                    # a skipped unit gets its cbf flags cleared, nothing else is written
            cu = work["cu"][:n]
            live = ~((cu["pred_mode"] == 1) & (cu["inter_kind"] == S.INTER_SKIP))
            if ne == 5:
                live = np.repeat(live[:1], 4)
            a, b = a[live], b[live]
        assert np.array_equal(a, b), (tag, f, a.tolist(), b.tolist())
    gy, wy = got["coeff_y"].reshape(64, 64), want["coeff_y"].reshape(64, 64)
    for i in range(n):
        cu = work["cu"][i]
        x, y, s = int(cu["x"]), int(cu["y"]), int(cu["size"])
        if int(cu["pred_mode"]) == 1 and int(cu["inter_kind"]) == S.INTER_SKIP:
            continue      # a skipped unit writes no coefficients (the reference's buffer keeps what an earlier picture left there)
        assert np.array_equal(gy[y:y + s, x:x + s], wy[y:y + s, x:x + s]), (tag, "coeff_y", i)
        for p in ("coeff_cb", "coeff_cr"):
            a, b = got[p].reshape(32, 32), want[p].reshape(32, 32)
            assert np.array_equal(a[y // 2:(y + s) // 2, x // 2:(x + s) // 2], b[y // 2:(y + s) // 2, x // 2:(x + s) // 2]), (tag, p, i)
    if not rec:
        return
    lw, lh = min(64, w - int(work["lcu_x"])), min(64, h - int(work["lcu_y"]))
    assert np.array_equal(got["rec_y"].reshape(64, 64)[:lh, :lw], want["rec_y"].reshape(64, 64)[:lh, :lw]), (tag, "rec_y")
    for p in ("rec_cb", "rec_cr"):
        assert np.array_equal(got[p].reshape(32, 32)[:lh // 2, :lw // 2], want[p].reshape(32, 32)[:lh // 2, :lw // 2]), (tag, p)


def test_have_cases():
    assert len(CASES) >= 6 and sum("i10_" in c for c in CASES) >= 3


@pytest.mark.parametrize("name", CASES)
def test_fixture_is_what_the_contract_says(name):
    g, w, h = load_case(name)
    if is16(g):
        assert g["work"].dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize == 13872 and g["result"].dtype.itemsize == S.LCU_RESULT16_DTYPE.itemsize == 25344
        assert int(g["work"]["src_y"].max()) > 255      # really 10-bit samples
    else:
        assert g["work"].dtype.itemsize == S.LCU_WORK_DTYPE.itemsize == 7728 and g["result"].dtype.itemsize == S.LCU_RESULT_DTYPE.itemsize == 19200
    nl = S.lcu_count(w, h)
    assert len(g["work"]) % nl == 0
    for wk in g["work"]:
        n = int(wk["num_cus"])
        cu = wk["cu"][:n]
        assert n >= 1 and (cu["pred_mode"] == 2).all() and np.isin(cu["size"], (8, 16, 32)).all()
        # the units tile the part of the LCU that is inside the picture
        cov = np.zeros((64, 64), np.int32)
        for c in cu:
            cov[c["y"]:c["y"] + c["size"], c["x"]:c["x"] + c["size"]] += 1
        lw, lh = min(64, w - int(wk["lcu_x"])), min(64, h - int(wk["lcu_y"]))
        assert (cov[:lh, :lw] == 1).all() and cov.sum() == lw * lh


@pytest.mark.parametrize("name", CASES)
def test_encode_lcu_oracle_matches_reference(oracle, name):
    g, w, h = load_case(name)
    pm = bool(g["work"]["pm_core"].any())        # PM-core presets: the quantiser reads the picture's rate tables
    if pm:
        fn = inter_oracle_fn(oracle, is16(g))
    else:
        fn = oracle.svt_oracle_encode_lcu16 if is16(g) else oracle.svt_oracle_encode_lcu
        fn.restype = None
        fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if is16(g) else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    pitches = (w + 32, w // 2 + 16, w // 2 + 16)
    pb = (C.c_uint32 * 3)(*pitches)
    for first in range(0, len(g["work"]), nl):
        # poisoned picture: a sample the restatement may not read yet shows up as a mismatch
        rec = [np.full((hh, p), 0xA5, sdt) for hh, p in zip((h, h // 2, h // 2), pitches)]
        mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
        rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
        for k in range(first, first + nl):
            work = np.ascontiguousarray(g["work"][k:k + 1])
            got = np.zeros(1, rdt)
            if pm:
                cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(int(g["picture_number"][k]))])
                fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, None, None, cost.ctypes.data, work.ctypes.data, got.ctypes.data)
            else:
                fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, work.ctypes.data, got.ctypes.data)
            compare_lcu(work[0], g["result"][k], got[0], w, h, (name, int(g["picture_number"][k]), int(g["lcu_index"][k])))


def ref_pictures(g, wide):
    """SvtAmdRefPicture records (host pointers) of the fixture's reference pictures by POC, and the arrays that back them"""
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    keep, out = [], {}
    for i, poc in enumerate(g["ref_pocs"].tolist()):
        planes = [np.ascontiguousarray(g[k][i]) for k in ("ref_y", "ref_cb", "ref_cr")]
        assert planes[0].dtype.itemsize == (2 if wide else 1)
        keep.append(planes)
        out[poc] = S.RefPicture(planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, sy, sc, ox, oy, rw, rh)
    return out, keep


def inter_oracle_fn(oracle, wide):
    fn = oracle.svt_oracle_encode_lcu_inter16 if wide else oracle.svt_oracle_encode_lcu_inter
    fn.restype = None
    fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    return fn


def test_have_inter_cases():
    assert len(INTER_CASES) >= 5
    kinds, dirs, sizes = set(), set(), set()
    for name in INTER_CASES:
        g, _, _ = load_case(name)
        for wk in g["work"]:
            cu = wk["cu"][:int(wk["num_cus"])]
            it = cu[cu["pred_mode"] == 1]
            kinds.update(it["inter_kind"].tolist()), dirs.update(it["inter_dir"].tolist()), sizes.update(it["size"].tolist())
    assert kinds == {S.INTER_AMVP, S.INTER_MERGE, S.INTER_SKIP} and {0, 2} <= dirs and sizes == {8, 16, 32, 64}


@pytest.mark.parametrize("name", INTER_CASES)
def test_encode_lcu_oracle_matches_reference_on_p_and_b_pictures(oracle, name):
    """every LCU of the recorded P / B pictures in raster order: inter units (prediction from the recorded reference pictures, encode
    loop, luma cbf decision of AMVP units, skipped units) and the intra units between them"""
    g, w, h = load_case(name)
    wide = is16(g)
    fn = inter_oracle_fn(oracle, wide)
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    refs, keep = ref_pictures(g, wide)
    nl = S.lcu_count(w, h)
    pitches = (w + 32, w // 2 + 16, w // 2 + 16)
    pb = (C.c_uint32 * 3)(*pitches)
    assert len(g["work"]) % nl == 0
    decided = 0
    for first in range(0, len(g["work"]), nl):
        rec = [np.full((hh, p), 0xA5, sdt) for hh, p in zip((h, h // 2, h // 2), pitches)]
        mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
        rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
        pic = int(g["picture_number"][first])
        cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(pic)])
        assert cost.size == 1560
        for k in range(first, first + nl):
            assert int(g["picture_number"][k]) == pic and int(g["lcu_index"][k]) == k - first
            work = np.ascontiguousarray(g["work"][k:k + 1])
            r0, r1 = (refs.get(int(v)) for v in g["ref_poc"][k])
            got = np.zeros(1, rdt)
            fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None, cost.ctypes.data,
               work.ctypes.data, got.ctypes.data)
            # LCUs of non-reference pictures the reference did not reconstruct (doRecon == 0): flags and coefficients only
            compare_lcu(work[0], g["result"][k], got[0], w, h, (name, pic, k - first), rec=not (int(g["dlf_off"][k]) & 2))
            cu = work[0]["cu"][:int(work[0]["num_cus"])]
            amvp = (cu["pred_mode"] == 1) & (cu["inter_kind"] == S.INTER_AMVP)
            decided += int(((g["result"][k]["cu"]["nz"][:len(cu), 0] != 0) & (g["result"][k]["cu"]["cbf"][:len(cu), 0] == 0) & amvp).sum())
    assert decided >= 0


def deblock_maps(works, results, w, h):
    """the three picture-level inputs of the deblocking kernels from the contract records: coding-unit map per 8x8 block, luma cbf
    per 4x4 block, QP per 8x8 block, tile-edge flags per LCU (svt-hevc_amd/csrc/encdec_kernels.hip:picture_deblock does the same)"""
    cu_dt = np.dtype([("mode", "u1"), ("dir", "u1"), ("size_log2", "u1"), ("pad", "u1"), ("mv", "<i2", (2, 2))])
    cumap = np.zeros((h // 8, w // 8), cu_dt)
    cbf = np.zeros((h // 4, w // 4), np.uint8)
    qp = np.zeros((h // 8, w // 8), np.uint8)
    edge = np.zeros(len(works), np.uint8)
    for i, (wk, rs) in enumerate(zip(works, results)):
        edge[i] = (1 if wk["tile_left"] else 0) | (2 if wk["tile_top"] else 0)
        for c in range(int(wk["num_cus"])):
            u = wk["cu"][c]
            x0, y0, n = int(wk["lcu_x"]) + int(u["x"]), int(wk["lcu_y"]) + int(u["y"]), int(u["size"])
            cumap["mode"][y0 // 8:(y0 + n) // 8, x0 // 8:(x0 + n) // 8] = u["pred_mode"]
            cumap["size_log2"][y0 // 8:(y0 + n) // 8, x0 // 8:(x0 + n) // 8] = n.bit_length() - 1
            if int(u["pred_mode"]) == 1:
                cumap["dir"][y0 // 8:(y0 + n) // 8, x0 // 8:(x0 + n) // 8] = u["inter_dir"]
                cumap["mv"][y0 // 8:(y0 + n) // 8, x0 // 8:(x0 + n) // 8] = u["mv"]
            qp[y0 // 8:(y0 + n) // 8, x0 // 8:(x0 + n) // 8] = u["qp"]
            if n == 64:      # four 32x32 transform units, result entries 1..4
                for t in range(4):
                    tx, ty = x0 + 32 * (t & 1), y0 + 32 * (t >> 1)
                    cbf[ty // 4:ty // 4 + 8, tx // 4:tx // 4 + 8] = rs["cu"]["cbf"][c + 1 + t][0]
            else:
                cbf[y0 // 4:(y0 + n) // 4, x0 // 4:(x0 + n) // 4] = rs["cu"]["cbf"][c][0]
    return cumap, cbf, qp, edge


def test_have_dlf_cases():
    assert len(DLF_CASES) >= 5 and sum("ref_pocs" in load_case(c)[0] for c in DLF_CASES) >= 2


def allows_mismatch(g, w, h, work0):
    """contextPtr->allowEncDecMismatch (Codec/EbEncDecProcess.c:2036-2054): in temporal layers > 0 at encMode >= 8 - and at encMode 7 in 4K - the
    encoder neither deblocks (EbCodingLoop.c:3081) nor applies SAO (EbEncDecProcess.c:3085) on its side: its reconstruction / reference
    picture is the picture as encoded (the SAO parameters are still decided, on that picture, and signalled)"""
    args = g["enc_args"].tolist()
    mode = int(args[args.index("-encMode") + 1])
    return int(work0["temporal_layer"]) > 0 and (mode >= 8 or (mode == 7 and w * h > 1920 * 1080 * 2))


def encode_and_deblock_pictures(oracle, name, g, w, h):
    """the chain on the CPU, picture by picture: encode pass of every LCU (checked against the records), boundary strengths from the unit
    lists, picture deblocking.  Yields (picture number, first record, works, results, un-deblocked planes, deblocked planes)."""
    from test_oracle_dlf_golden import oracle_bs, oracle_dlf
    wide = is16(g)
    inter = "ref_pocs" in g       # P / B pictures: the (deblocked) reference pictures and rate tables come with the fixture
    fn = inter_oracle_fn(oracle, wide)
    refs, keep = ref_pictures(g, wide) if inter else ({}, None)
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    pitches = (w, w // 2, w // 2)
    pb = (C.c_uint32 * 3)(*pitches)
    for first in range(0, len(g["work"]), nl):
        f = int(g["picture_number"][first])     # the encoder's output is in display order
        rec = [np.zeros((hh, p), sdt) for hh, p in zip((h, h // 2, h // 2), pitches)]
        mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
        rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
        got = np.zeros(nl, rdt)
        r0, r1 = (refs.get(int(v)) for v in g["ref_poc"][first]) if inter else (None, None)
        cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)]) if inter and f in g["cost_pictures"].tolist() else None
        for k in range(nl):
            work = np.ascontiguousarray(g["work"][first + k:first + k + 1])
            fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None,
               cost.ctypes.data if cost is not None else None, work.ctypes.data, got[k:k + 1].ctypes.data)
            compare_lcu(work[0], g["result"][first + k], got[k], w, h, (name, f, k), rec=False)
        cumap, cbf, qp, edge = deblock_maps(g["work"][first:first + nl], got, w, h)
        hdr = dict(width=w, height=h, bytes_per_sample=2 if wide else 1, qp_stride=w // 8, tc_offset=0, beta_offset=0, cb_qp_offset=0,
                   cr_qp_offset=0, slice_type=int(g["work"][first]["slice_type"]))
        pic = dict(hdr=hdr, cumap=cumap.reshape(-1), cbf=cbf.reshape(-1),
                   refpoc=np.ascontiguousarray(g["ref_poc"][first]) if inter else np.zeros(2, np.uint64), lcu_edge=edge,
                   bsv=np.zeros((nl, 256), np.uint8), bsh=np.zeros((nl, 256), np.uint8))
        pic["bsv"], pic["bsh"] = oracle_bs(oracle, pic)
        pic["pre"], pic["qp"] = rec, qp.reshape(-1)
        fin = [r.copy() for r in rec] if allows_mismatch(g, w, h, g["work"][first]) else oracle_dlf(oracle, pic)
        yield f, first, g["work"][first:first + nl], got, rec, fin


@pytest.mark.parametrize("name", DLF_CASES)
def test_encode_then_deblock_oracle_matches_the_encoders_output(oracle, name):
    """the chain the device runs - encode pass of every LCU, boundary strengths from the unit lists, picture deblocking - restated
    on the CPU from the three pinned oracle pieces, against the reference encoder's own reconstruction output (SAO off)"""
    g, w, h = load_case(name)
    for f, first, works, got, pre, out in encode_and_deblock_pictures(oracle, name, g, w, h):
        for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
            bad = np.argwhere(out[p] != g[nm][f])
            assert len(bad) == 0, (name, f, nm, len(bad), bad[:4].tolist())


def encoder_order_lcu(pre, fin, p, x0, y0, lw, lh, w, h, right_follows=True, below_follows=True):
    """The samples of an LCU as the reference's SAO decision sees them (Codec/EbCodingLoop.c:4600-4750: the LCU's own three deblocking
    drivers have run, those of the LCUs to its right and below have not): the deblocked picture, except the last 4 columns / rows of
    the LCU (in the plane's own samples) where a neighbour still follows - those belong to 8x8 filter blocks centred on the LCU
    boundary (LCUBoundaryDLFCore of the NEXT LCU, EbDeblockingFilter.c:2828) and still hold un-deblocked samples."""
    sh = 1 if p else 0
    bx, by, bw, bh = x0 >> sh, y0 >> sh, lw >> sh, lh >> sh
    blk = fin[p][by:by + bh, bx:bx + bw].copy()
    if x0 + lw < w and right_follows:
        blk[:, bw - 4:] = pre[p][by:by + bh, bx + bw - 4:bx + bw]
    if y0 + lh < h and below_follows:
        blk[bh - 4:, :] = pre[p][by + bh - 4:by + bh, bx:bx + bw]
    return blk


def test_have_sao_cases():
    assert len(SAO_CASES) >= 4


@pytest.mark.parametrize("name", SAO_CASES)
def test_encoder_order_sao_statistics_from_two_pictures(oracle, name):
    """what the reference's SaoGenerationDecision gathered for every LCU (recorded inside the encoder) = statistics of the composite of the
    un-deblocked and the finished deblocked picture: no LCU-by-LCU deblocking order is needed to reproduce them"""
    g, w, h = load_case(name)
    wide = is16(g)
    vp, u32 = C.c_void_p, C.c_uint32
    oracle.svt_oracle_GatherSaoStatistics.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    oracle.svt_oracle_GatherSaoStatistics.restype = None
    sao = g["sao"]
    checked = 0
    for f, first, works, got, pre, fin in encode_and_deblock_pictures(oracle, name, g, w, h):
        for k, wk in enumerate(works):
            x0, y0 = int(wk["lcu_x"]), int(wk["lcu_y"])
            sel = sao[(sao["picture_number"] == f) & (sao["origin_x"] == x0) & (sao["origin_y"] == y0)]
            if not len(sel):
                continue          # the encode pass shut SAO off for this LCU (EbCodingLoop.c:4678-4707)
            r = sel[0]
            lw, lh = min(64, w - x0), min(64, h - y0)
            ncomp = 3 if r["mm_sao"] else (1 if r["temporal_layer"] < 2 else 0)
            for p in range(ncomp):
                sh = 1 if p else 0
                cols_ = (w + 63) // 64
                blk = np.ascontiguousarray(encoder_order_lcu(pre, fin, p, x0, y0, lw, lh, w, h, not wk["tile_right"],
                                                             k + cols_ >= len(works) or not works[k + cols_]["tile_top"]))
                src = np.ascontiguousarray(wk[("src_y", "src_cb", "src_cr")[p]].reshape(64 >> sh, 64 >> sh))
                bd, bc = np.zeros(32, np.int32), np.zeros(32, np.uint16)
                ed, ec = np.zeros((4, 5), np.int32), np.zeros((4, 5), np.uint16)
                oracle.svt_oracle_GatherSaoStatistics(2 if wide else 1, 0 if r["mm_sao"] else 1, src.ctypes.data, 64 >> sh, blk.ctypes.data, blk.shape[1],
                                                      lw >> sh, lh >> sh, bd.ctypes.data, bc.ctypes.data, ed.ctypes.data, ec.ctypes.data)
                tag = (name, f, k, p)
                assert np.array_equal(ed, r["eo_diff"][p]) and np.array_equal(ec, r["eo_count"][p]), (tag, "eo", ed.tolist(), r["eo_diff"][p].tolist())
                if r["mm_sao"]:
                    assert np.array_equal(bd, r["bo_diff"][p]) and np.array_equal(bc, r["bo_count"][p]), (tag, "bo")
            checked += 1
    assert checked >= len(g["work"]) // 2, checked



def sao_inputs_of_picture(g, f, works, w, h):
    """decision inputs of one picture from the fixture's records: rate parameters, enable map (0 = the encode pass shut SAO off there),
    edge flags (1 / 4: no left / upper merge candidate; 2 / 8: right / bottom tile edge for the application) and the reference's decisions"""
    from test_oracle_saodec_golden import LCU, params_of
    sao = g["sao"]
    rr = sao[sao["picture_number"] == f]
    cols, rows = (w + 63) // 64, (h + 63) // 64
    enable, params, want = np.zeros(cols * rows, np.uint8), np.zeros(cols * rows, LCU), np.zeros(cols * rows, LCU)
    idx = (rr["origin_y"] // 64) * cols + rr["origin_x"] // 64
    enable[idx] = 1
    for k, wk in enumerate(works):
        below = k + cols
        params["edge_flags"][k] = (1 if wk["tile_left"] else 0) | (2 if wk["tile_right"] else 0) | (4 if wk["tile_top"] else 0) | \
                                  (8 if (below >= len(works) or works[below]["tile_top"]) else 0)
    for k in ("merge_left", "merge_up", "type", "offset", "band"):
        want[k][idx] = rr["out"][k]
    return (params_of(rr[0]) if len(rr) else None), enable, params, want, idx


@pytest.mark.parametrize("name", SAO_CASES)
def test_encode_deblock_sao_oracle_matches_the_encoders_output(oracle, name):
    """the whole closed-loop tail restated from pinned pieces and two pictures in memory: encode pass -> deblocking -> SAO statistics of the
    encoder-order composite -> parameter decision (merge wavefront) -> SAO application = the reference encoder's reconstruction output
    with every in-loop filter on; the decided parameters equal the encoder's own, LCU by LCU"""
    from test_oracle_dlf_golden import oracle_sao
    from test_oracle_saodec_golden import STATS, oracle_decide_picture, same_decision
    g, w, h = load_case(name)
    wide = is16(g)
    vp, u32 = C.c_void_p, C.c_uint32
    oracle.svt_oracle_GatherSaoStatistics.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    oracle.svt_oracle_GatherSaoStatistics.restype = None
    cols, rows = (w + 63) // 64, (h + 63) // 64
    for f, first, works, got, pre, fin in encode_and_deblock_pictures(oracle, name, g, w, h):
        P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
        if P is None:
            out = fin
        else:
            stats = np.zeros((3, cols * rows), STATS)
            ncomp = 3 if P["mm_sao"][0] else (1 if P["temporal_layer"][0] < 2 else 0)
            for k, wk in enumerate(works):
                x0, y0 = int(wk["lcu_x"]), int(wk["lcu_y"])
                lw, lh = min(64, w - x0), min(64, h - y0)
                for p in range(ncomp):
                    sh = 1 if p else 0
                    blk = np.ascontiguousarray(encoder_order_lcu(pre, fin, p, x0, y0, lw, lh, w, h, not wk["tile_right"],
                                                                 k + cols >= len(works) or not works[k + cols]["tile_top"]))
                    src = np.ascontiguousarray(wk[("src_y", "src_cb", "src_cr")[p]].reshape(64 >> sh, 64 >> sh))
                    st = stats[p][k:k + 1]
                    oracle.svt_oracle_GatherSaoStatistics(2 if wide else 1, 0 if P["mm_sao"][0] else 1, src.ctypes.data, 64 >> sh, blk.ctypes.data,
                                                          blk.shape[1], lw >> sh, lh >> sh, st["boDiff"].ctypes.data, st["boCount"].ctypes.data,
                                                          st["eoDiff"].ctypes.data, st["eoCount"].ctypes.data)
            dec, _ = oracle_decide_picture(oracle, dict(P=P, stats=stats, enable=enable, params=params, cols=cols, rows=rows))
            for i in idx:
                assert same_decision(dec[i], want[i]), (name, f, int(i), dec[i], want[i])
            dec["edge_flags"] = params["edge_flags"]
            out = fin if allows_mismatch(g, w, h, works[0]) else oracle_sao(oracle, fin, 2 if wide else 1, w, h, dec, 1, 1)
        for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
            bad = np.argwhere(out[p] != g[nm][f])
            assert len(bad) == 0, (name, f, nm, len(bad), bad[:4].tolist())
        # ... and, padded as PadRefAndSetFlags does (EbEncDecProcess.c:1805: edge replication), it IS the reference picture the later
        # pictures of the sequence predicted from
        if "ref_pocs" in g and f in g["ref_pocs"].tolist():
            sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
            i = g["ref_pocs"].tolist().index(f)
            for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):
                px, py = (ox >> 1, oy >> 1) if p else (ox, oy)
                padded = np.pad(out[p], ((py, py), (px, px)), mode="edge")
                assert padded.shape[1] == (sc if p else sy)
                assert np.array_equal(padded.reshape(-1), g[nm][i]), (name, f, nm)
