"""CPU-only: pins oracle/svt_oracle_fullloop.c:svt_oracle_pmcore_quantize (the encode pass's UnifiedQuantizeInvQuantize with
rdoqPmCoreMethod == EB_PMCORE: regular quantisation, for luma the 4x4-block re-decision among 100 / 70 / 50 % scalings by SSE +
lambda * rate, every level de-quantised again) against records of real calls of encMode 2 / 3 / 4 encodes
(tests/golden/uqiqpm_*.npz, made by tests/golden/make_uqiq_golden.py pm:<name>)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "uqiqpm_*.npz")))
UNIT = np.dtype([("size", "u1"), ("qp", "u1"), ("bit_depth", "u1"), ("slice_type", "u1"), ("component", "u1"), ("cand_type", "u1"),
                 ("pad", "u1", 2), ("lambda", "<u4")])


def load_uqiqpm_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "uqiqpm_%s.npz" % name)))
    g["offsets"] = np.concatenate([[0], np.cumsum(g["size"].astype(np.int64) ** 2)])
    return g


def unit_of(g, i):
    u = np.zeros(1, UNIT)
    for k in ("size", "qp", "bit_depth", "slice_type", "component", "cand_type", "lambda"):
        u[k] = g[k][i]
    return u


def blocks_of(g, i):
    a, b, n = int(g["offsets"][i]), int(g["offsets"][i + 1]), int(g["size"][i])
    return [np.ascontiguousarray(g[k][a:b]).reshape(n, n) for k in ("coeff", "quant", "recon")]


def test_layout_and_cases():
    assert UNIT.itemsize == 12 and len(CASES) == 3


@pytest.mark.parametrize("name", CASES)
def test_pmcore_quantize_oracle_matches_reference(oracle, name):
    g = load_uqiqpm_case(name)
    oracle.svt_oracle_pmcore_quantize.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_pmcore_quantize.restype = None
    changed = 0
    for i in range(len(g["size"])):
        coeff, wq, wr = blocks_of(g, i)
        q, r, nz, u = np.zeros_like(coeff), np.zeros_like(coeff), np.zeros(1, np.uint32), unit_of(g, i)
        cost = g["cost_tables"][int(g["cost_index"][i]):int(g["cost_index"][i]) + 1]
        oracle.svt_oracle_pmcore_quantize(cost.ctypes.data, u.ctypes.data, coeff.ctypes.data, q.ctypes.data, r.ctypes.data, nz.ctypes.data)
        assert np.array_equal(q, wq), (name, i, u, np.argwhere(q != wq)[:4].tolist())
        assert np.array_equal(r, wr) and int(nz[0]) == int(g["nz_out"][i]), (name, i)
        if int(g["component"][i]) == 0 and int(nz[0]):
            # would the plain quantiser have said something else?
            u2 = u.copy()
            u2["component"] = 1
            q2 = np.zeros_like(coeff)
            oracle.svt_oracle_pmcore_quantize(cost.ctypes.data, u2.ctypes.data, coeff.ctypes.data, q2.ctypes.data, r.ctypes.data, nz.ctypes.data)
            changed += not np.array_equal(q, q2)
    assert changed >= (1 if "noise" in name else 10), changed
