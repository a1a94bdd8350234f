"""CPU-only: pins oracle/svt_oracle_txfm.c:svt_oracle_unified_quantize against records of real UnifiedQuantizeInvQuantize calls
of the reference's encode pass (tests/golden/uqiq_*.npz, made by tests/golden/make_uqiq_golden.py), and checks its optional
branches (DC-only shape, partial shapes, dead-zone override, isolated-coefficient clean-up, contouring / forced cbf) against
the reference leaf functions they are made of."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "uqiq_*.npz")))
UNIT = np.dtype([("size", "u1"), ("qp", "u1"), ("bit_depth", "u1"), ("slice_type", "u1"), ("shape", "u1"), ("clean_sparse", "u1"),
                 ("enable_cb_flag", "u1"), ("contouring_flag", "u1"), ("component", "u1"), ("temporal_layer", "u1"), ("pad", "u1", 2),
                 ("dz_offset", "<u4")])


def load_uqiq_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "uqiq_%s.npz" % name)))
    g["offsets"] = np.concatenate([[0], np.cumsum(g["size"].astype(np.int64) ** 2)])
    return g


def unit_of(g, i):
    u = np.zeros(1, UNIT)
    for k in ("size", "qp", "bit_depth", "slice_type", "shape", "clean_sparse", "enable_cb_flag", "contouring_flag", "component",
              "temporal_layer", "dz_offset"):
        u[k] = g[k][i]
    return u


def blocks_of(g, i):
    a, b, n = int(g["offsets"][i]), int(g["offsets"][i + 1]), int(g["size"][i])
    return [np.ascontiguousarray(g[k][a:b]).reshape(n, n) for k in ("coeff", "quant_in", "recon_in", "quant", "recon")]


def oracle_call(oracle, u, coeff, quant, recon):
    oracle.svt_oracle_unified_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    oracle.svt_oracle_unified_quantize.restype = None
    nz = C.c_uint32(0)
    oracle.svt_oracle_unified_quantize(u.ctypes.data, coeff.ctypes.data, coeff.shape[1], quant.ctypes.data, recon.ctypes.data, C.byref(nz))
    return nz.value


def test_unit_size():
    assert UNIT.itemsize == 16


def test_have_cases():
    assert len(CASES) >= 5


@pytest.mark.parametrize("name", CASES)
def test_uqiq_oracle_matches_reference(oracle, name):
    g = load_uqiq_case(name)
    for i in range(len(g["size"])):
        coeff, qin, rin, wq, wr = blocks_of(g, i)
        quant, recon = qin.copy(), rin.copy()
        nz = oracle_call(oracle, unit_of(g, i), coeff, quant, recon)
        assert nz == int(g["nz_out"][i]), (name, i)
        assert np.array_equal(quant, wq) and np.array_equal(recon, wr), (name, i, int(g["size"][i]))


def random_units(rng, n):
    u = np.zeros(n, UNIT)
    u["size"] = rng.choice([4, 8, 16, 32], n)
    u["qp"], u["bit_depth"] = rng.integers(0, 52, n), rng.choice([8, 10], n)
    u["slice_type"] = rng.integers(0, 3, n)
    u["shape"] = np.where(u["size"] >= 16, rng.integers(0, 4, n), np.where(u["size"] == 8, rng.choice([0, 1, 3], n), rng.choice([0, 3], n)))
    for f in ("clean_sparse", "enable_cb_flag", "contouring_flag"):
        u[f] = rng.integers(0, 2, n)
    u["component"], u["temporal_layer"] = rng.integers(0, 3, n), rng.integers(0, 3, n)
    u["dz_offset"] = np.where(rng.random(n) < 0.3, rng.integers(1, 20, n), 0)
    return u


def random_coeff(rng, size, k):
    amp = [2, 40, 800, 32767][k % 4]
    c = rng.integers(-amp, amp + 1, (size, size))
    if k % 3 == 0:   # sparse: isolated coefficients for the clean-up branch
        c = c * (rng.random((size, size)) < 0.04)
    return c.astype(np.int16)


@pytest.mark.skipif(not os.path.exists(S.REF_SO), reason="oracle/_ref/libsvtref.so not built (needs /root/reference)")
def test_uqiq_optional_branches_match_reference_function(oracle):
    """shapes, dead-zone override, isolated-coefficient clean-up, contouring and forced cbf: the reference function itself,
    called with stand-in context objects (oracle/ref_harness_uqiq_dump.c:svt_ref_unified_quantize)"""
    ref = C.CDLL(S.REF_SO)
    ref.svt_ref_unified_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    ref.svt_ref_unified_quantize.restype = None
    rng = np.random.default_rng(12)
    units = random_units(rng, 1500)
    hit = {"dc": 0, "clean": 0, "contour": 0, "cbf": 0, "dz": 0}
    for k in range(len(units)):
        u = units[k:k + 1]
        n = int(u["size"][0])
        coeff = random_coeff(rng, n, k)
        q0, r0 = rng.integers(-9, 9, (n, n)).astype(np.int16), rng.integers(-9, 9, (n, n)).astype(np.int16)
        wq, wr, gq, gr = q0.copy(), r0.copy(), q0.copy(), r0.copy()
        wnz, coeff_ref = C.c_uint32(0), coeff.copy()
        ref.svt_ref_unified_quantize(u.ctypes.data, coeff_ref.ctypes.data, n, wq.ctypes.data, wr.ctypes.data, C.byref(wnz))
        gnz = oracle_call(oracle, u, coeff, gq, gr)
        assert gnz == wnz.value and np.array_equal(gq, wq) and np.array_equal(gr, wr), (k, u)
        hit["dc"] += int(u["shape"][0] == 3)
        hit["dz"] += int(u["dz_offset"][0] != 0)
    assert hit["dc"] > 50 and hit["dz"] > 100
