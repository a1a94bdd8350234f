"""CPU-only: pins oracle/svt_oracle_encodepass.c:svt_oracle_encode_lcu (the coding-unit loop of EncodePass for LCUs of intra 2Nx2N
units, neighbours read from the un-deblocked picture + a mode map) against records of real EncodePass calls of the reference
(tests/golden/encodepass_*.npz, made by tests/golden/make_encodepass_golden.py with the loop filters off)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "encodepass_*.npz")))


def load_case(name):
    g = np.load(os.path.join(S.GOLDEN_DIR, "encodepass_%s.npz" % name))
    w, h = int(g["clip"][1]), int(g["clip"][2])
    return g, w, h


def is16(g):
    """10-bit encodes (EncodePass with is16bit) are recorded with 16-bit source / reconstruction samples"""
    return g["work"].dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize


def compare_lcu(work, want, got, w, h, tag):
    """cbf / DC-only / counts of the units, the quantised coefficients of every unit area and the LCU's reconstruction inside the picture"""
    n = int(work["num_cus"])
    for f in ("cbf", "only_dc", "nz"):
        assert np.array_equal(got["cu"][f][:n], want["cu"][f][:n]), (tag, f, got["cu"][f][:n].tolist(), want["cu"][f][:n].tolist())
    gy, wy = got["coeff_y"].reshape(64, 64), want["coeff_y"].reshape(64, 64)
    for i in range(n):
        cu = work["cu"][i]
        x, y, s = int(cu["x"]), int(cu["y"]), int(cu["size"])
        assert np.array_equal(gy[y:y + s, x:x + s], wy[y:y + s, x:x + s]), (tag, "coeff_y", i)
        for p in ("coeff_cb", "coeff_cr"):
            a, b = got[p].reshape(32, 32), want[p].reshape(32, 32)
            assert np.array_equal(a[y // 2:(y + s) // 2, x // 2:(x + s) // 2], b[y // 2:(y + s) // 2, x // 2:(x + s) // 2]), (tag, p, i)
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
        assert g["work"].dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize == 13328 and g["result"].dtype.itemsize == S.LCU_RESULT16_DTYPE.itemsize == 25344
        assert int(g["work"]["src_y"].max()) > 255      # really 10-bit samples
    else:
        assert g["work"].dtype.itemsize == S.LCU_WORK_DTYPE.itemsize == 7184 and g["result"].dtype.itemsize == S.LCU_RESULT_DTYPE.itemsize == 19200
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
