"""CPU-only: pins oracle/svt_oracle_fullloop.c (and with it the transform / quantiser / distortion / rate leaf
restatements it is composed of) against records of real ProductFullLoop calls made by the reference's mode decision
(tests/golden/fullloop_*.npz, made by tests/golden/make_fullloop_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[9:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "fullloop_*.npz")))


class FullLoopIn(C.Structure):
    _fields_ = [("size", C.c_uint32), ("qp", C.c_uint32), ("slice_type", C.c_uint32), ("pf_mode", C.c_uint32),
                ("cand_type", C.c_uint32), ("intra_luma_mode", C.c_uint32), ("full_lambda", C.c_uint32),
                ("cbf_bits", C.c_uint32 * 4), ("ycbf", C.c_uint32), ("coeff_bits", C.c_uint64), ("dist", C.c_uint64 * 2)]


class FullLoopOut(C.Structure):
    _fields_ = [("nz", C.c_uint32 * 5), ("ycbf", C.c_uint32), ("coeff_bits", C.c_uint64), ("dist", C.c_uint64 * 2),
                ("ydc", C.c_int16 * 4), ("cand_nz", C.c_uint16 * 4)]


def load_fullloop_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "fullloop_%s.npz" % name)))
    off = np.concatenate([[0], np.cumsum(g["size"].astype(np.int64) ** 2)])
    g["offsets"] = off
    return g


def record_in(g, i):
    fin = FullLoopIn()
    fin.size, fin.qp, fin.slice_type, fin.pf_mode = int(g["size"][i]), int(g["qp"][i]), int(g["slice_type"][i]), int(g["pf_mode"][i]) | (int(g["pm_core"][i]) << 16 if "pm_core" in g else 0)
    fin.cand_type, fin.intra_luma_mode, fin.full_lambda = int(g["cand_type"][i]), int(g["intra_luma_mode"][i]), int(g["full_lambda"][i])
    for k in range(4):
        fin.cbf_bits[k] = int(g["cbf_bits"][i][k])
    fin.ycbf, fin.coeff_bits = int(g["ycbf_before"][i]), int(g["bits_in"][i])
    fin.dist[0], fin.dist[1] = int(g["dist_in"][i][0]), int(g["dist_in"][i][1])
    return fin


def check_out(g, i, out, quant, recon, what):
    a, b = int(g["offsets"][i]), int(g["offsets"][i + 1])
    size, pf = int(g["size"][i]), int(g["pf_mode"][i])
    wq, wr = g["quant"][a:b].reshape(size, size), g["recon"][a:b].reshape(size, size)
    # the reference quantises only the (T >> pf) x (T >> pf) area of every transform unit; the rest of its buffers is
    # stale, so only the area is compared
    T = 32 if size == 64 else size
    ar = T >> pf
    for ty in range(0, size, T):
        for tx in range(0, size, T):
            assert np.array_equal(quant[ty:ty + ar, tx:tx + ar], wq[ty:ty + ar, tx:tx + ar]), (what, i, "quant")
            assert np.array_equal(recon[ty:ty + ar, tx:tx + ar], wr[ty:ty + ar, tx:tx + ar]), (what, i, "recon")
    idx = range(1, 5) if size == 64 else range(0, 1)
    for k in idx:
        assert out.nz[k] == int(g["nz_out"][i][k]), (what, i, "nz", k)
    assert out.ycbf == int(g["ycbf_after"][i]), (what, i, "ycbf")
    assert out.coeff_bits == int(g["bits_out"][i]), (what, i, "bits", out.coeff_bits, int(g["bits_out"][i]))
    assert (out.dist[0], out.dist[1]) == (int(g["dist_out"][i][0]), int(g["dist_out"][i][1])), (what, i, "dist")
    n = 4 if size == 64 else 1
    assert list(out.ydc)[:n] == g["ydc"][i][:n].tolist() and list(out.cand_nz)[:n] == g["cand_nz"][i][:n].tolist(), (what, i)


def test_have_cases():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_fullloop_oracle_matches_reference(oracle, name):
    g = load_fullloop_case(name)
    oracle.svt_oracle_product_full_loop_luma.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_product_full_loop_luma.restype = None
    oracle.svt_oracle_product_full_loop_luma_cabac.argtypes = [C.c_void_p] * 7
    oracle.svt_oracle_product_full_loop_luma_cabac.restype = None
    seen = set()
    n_cabac = 0
    for i in range(len(g["size"])):
        a, b = int(g["offsets"][i]), int(g["offsets"][i + 1])
        size = int(g["size"][i])
        res = np.ascontiguousarray(g["residual"][a:b]).reshape(size, size)
        quant, recon = res.copy(), np.zeros_like(res)
        fin, out = record_in(g, i), FullLoopOut()
        cost = np.ascontiguousarray(g["cost"][i:i + 1])
        if "cabac_update" in g and g["cabac_update"][i]:
            # coeffCabacUpdate: the candidate's context model goes in and comes out (bits AND states are pinned)
            model = np.ascontiguousarray(g["ctx_in"][i]).copy()
            oracle.svt_oracle_product_full_loop_luma_cabac(cost.ctypes.data, C.addressof(fin), res.ctypes.data, quant.ctypes.data,
                                                           recon.ctypes.data, model.ctypes.data, C.addressof(out))
            assert np.array_equal(model, g["ctx_out"][i]), (name, i, "context model")
            n_cabac += 1
        else:
            oracle.svt_oracle_product_full_loop_luma(cost.ctypes.data, C.addressof(fin), res.ctypes.data, quant.ctypes.data,
                                                     recon.ctypes.data, C.addressof(out))
        check_out(g, i, out, quant, recon, name)
        seen.add((size, int(g["cand_type"][i]), int(g["pf_mode"][i]), int(g["nz_out"][i].sum() > 0)))
    assert len(seen) >= 3
    assert (n_cabac > 0) == name.startswith("cabac")
