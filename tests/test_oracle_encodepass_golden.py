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
CASES = [c for c in ALL if not c.startswith("dlf_") and c not in INTER_CASES]  # all-intra, loop filters off: per-LCU reconstruction comparable
DLF_CASES = [c for c in ALL if c.startswith("dlf_")]        # deblocking on, SAO off: the encoder's output picture is the reference


def load_case(name):
    g = np.load(os.path.join(S.GOLDEN_DIR, "encodepass_%s.npz" % name))
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


@pytest.mark.parametrize("name", DLF_CASES)
def test_encode_then_deblock_oracle_matches_the_encoders_output(oracle, name):
    """the chain the device runs - encode pass of every LCU, boundary strengths from the unit lists, picture deblocking - restated
    on the CPU from the three pinned oracle pieces, against the reference encoder's own reconstruction output (SAO off)"""
    from test_oracle_dlf_golden import oracle_bs, oracle_dlf
    g, w, h = load_case(name)
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
        cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)]) if inter else None
        for k in range(nl):
            work = np.ascontiguousarray(g["work"][first + k:first + k + 1])
            fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None,
               cost.ctypes.data if inter else None, work.ctypes.data, got[k:k + 1].ctypes.data)
            compare_lcu(work[0], g["result"][first + k], got[k], w, h, (name, f, k), rec=False)
        cumap, cbf, qp, edge = deblock_maps(g["work"][first:first + nl], got, w, h)
        hdr = dict(width=w, height=h, bytes_per_sample=2 if wide else 1, qp_stride=w // 8, tc_offset=0, beta_offset=0, cb_qp_offset=0,
                   cr_qp_offset=0, slice_type=int(g["work"][first]["slice_type"]))
        pic = dict(hdr=hdr, cumap=cumap.reshape(-1), cbf=cbf.reshape(-1),
                   refpoc=np.ascontiguousarray(g["ref_poc"][first]) if inter else np.zeros(2, np.uint64), lcu_edge=edge,
                   bsv=np.zeros((nl, 256), np.uint8), bsh=np.zeros((nl, 256), np.uint8))
        pic["bsv"], pic["bsh"] = oracle_bs(oracle, pic)
        pic["pre"], pic["qp"] = rec, qp.reshape(-1)
        out = oracle_dlf(oracle, pic)
        for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
            bad = np.argwhere(out[p] != g[nm][f])
            assert len(bad) == 0, (name, f, nm, len(bad), bad[:4].tolist())
