"""CPU-only: pins the chroma composite of oracle/svt_oracle_fullloop.c against records of real FullLoop_R +
CuFullDistortionFastTuMode_R call pairs made by the reference's mode decision (tests/golden/chromaloop_*.npz, made by
tests/golden/make_chromaloop_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "chromaloop_*.npz")))


class ChromaLoopIn(C.Structure):
    _fields_ = [("size", C.c_uint32), ("cb_qp", C.c_uint32), ("cr_qp", C.c_uint32), ("slice_type", C.c_uint32),
                ("pf_mode", C.c_uint32), ("cand_type", C.c_uint32), ("intra_luma_mode", C.c_uint32), ("pad", C.c_uint32)]


class ChromaLoopOut(C.Structure):
    _fields_ = [("nz", (C.c_uint32 * 5) * 2), ("cbf", C.c_uint32 * 2), ("coeff_bits", C.c_uint64 * 2),
                ("dist", (C.c_uint64 * 2) * 2)]


def load_chromaloop_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "chromaloop_%s.npz" % name)))
    g["offsets"] = np.concatenate([[0], np.cumsum(2 * (g["size"].astype(np.int64) // 2) ** 2)])
    return g


def record_in(g, i):
    fin = ChromaLoopIn()
    fin.size, fin.cb_qp, fin.cr_qp = int(g["size"][i]), int(g["cb_qp"][i]), int(g["cr_qp"][i])
    fin.slice_type, fin.pf_mode = int(g["slice_type"][i]), int(g["pf_mode"][i])
    fin.cand_type, fin.intra_luma_mode = int(g["cand_type"][i]), int(g["intra_luma_mode"][i])
    return fin


def record_planes(g, i, key):
    """[Cb, Cr] planes of record i, each (size/2, size/2)"""
    a, b = int(g["offsets"][i]), int(g["offsets"][i + 1])
    c = int(g["size"][i]) // 2
    return np.ascontiguousarray(g[key][a:b]).reshape(2, c, c)


def check_out(g, i, out, quant, recon, what):
    size, pf = int(g["size"][i]), int(g["pf_mode"][i])
    c = size // 2
    wq, wr = record_planes(g, i, "quant"), record_planes(g, i, "recon")
    T = 16 if size == 64 else c
    ar = T >> (0 if T == 4 else (1 if (T == 8 and pf == 2) else pf))
    # the caller hands in zeros for the accumulators (EbProductCodingLoop.c:4250-4258, :4476-4483)
    assert not g["cbf_in"][i].any() and not g["bits_in"][i].any() and not g["dist_in"][i].any()
    for p in range(2):
        for ty in range(0, c, T):
            for tx in range(0, c, T):
                assert np.array_equal(quant[p][ty:ty + ar, tx:tx + ar], wq[p][ty:ty + ar, tx:tx + ar]), (what, i, p, "quant")
                assert np.array_equal(recon[p][ty:ty + ar, tx:tx + ar], wr[p][ty:ty + ar, tx:tx + ar]), (what, i, p, "recon")
        for k in (range(1, 5) if size == 64 else range(0, 1)):
            assert out.nz[p][k] == int(g["nz_out"][i][p][k]), (what, i, p, "nz", k)
        assert out.cbf[p] == int(g["cbf_out"][i][p]), (what, i, p, "cbf")
        assert out.coeff_bits[p] == int(g["bits_out"][i][p]), (what, i, p, "bits", out.coeff_bits[p], int(g["bits_out"][i][p]))
        assert (out.dist[p][0], out.dist[p][1]) == tuple(int(v) for v in g["dist_out"][i][p]), (what, i, p, "dist")


def test_have_cases():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_chromaloop_oracle_matches_reference(oracle, name):
    g = load_chromaloop_case(name)
    oracle.svt_oracle_full_loop_chroma.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_full_loop_chroma.restype = None
    seen = set()
    for i in range(len(g["size"])):
        res = record_planes(g, i, "residual")
        quant, recon = res.copy(), np.zeros_like(res)
        fin, out = record_in(g, i), ChromaLoopOut()
        cost = np.ascontiguousarray(g["cost"][i:i + 1])
        ptrs = [(C.c_void_p * 2)(a[0].ctypes.data, a[1].ctypes.data) for a in (res, quant, recon)]
        oracle.svt_oracle_full_loop_chroma(cost.ctypes.data, C.addressof(fin), ptrs[0], ptrs[1], ptrs[2], C.addressof(out))
        check_out(g, i, out, quant, recon, name)
        seen.add((int(g["size"][i]), int(g["cand_type"][i]), int(g["pf_mode"][i]), int(g["nz_out"][i].sum() > 0)))
    assert len(seen) >= 3
