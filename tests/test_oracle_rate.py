"""CPU-only: oracle/svt_oracle_rate.c against the reference's EstimateQuantizedCoefficients_Lossy
(Codec/EbCoeffEstimation_Intrinsic.c:1415) called in oracle/_ref/libsvtref.so, with cost tables produced by the
reference's own PrecomputeCabacCost (:161) from random CABAC context states."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
u32, vp, u64 = C.c_uint32, C.c_void_p, C.c_uint64

COST = np.dtype([("last", "<u4", 176), ("sig", "u1", 84), ("g1", "u1", 48), ("g2", "u1", 12), ("sigml", "u1", 8),
                 ("g1x", "<u2", 96), ("sigv", "u1", (32, 16))])


def make_cost(seed):
    """CabacCost_t filled by the reference from a random context-model state; CabacBitsLast (not written by
    PrecomputeCabacCost's coefficient part for every entry) is randomised as well."""
    rng = np.random.default_rng(seed)
    cost = np.zeros(1, COST)
    cost["last"] = rng.integers(0, 4000, 176)
    # EB_ContextModel is a 32-bit word holding a 7-bit state; the buffer is far larger than CabacEncodeContext_t
    # and is only read
    ctx = rng.integers(0, 126, 1 << 14).astype(np.uint32)
    ref.PrecomputeCabacCost(vp(cost.ctypes.data), vp(ctx.ctypes.data))
    return cost


def random_tu(rng, size, density, big):
    c = np.zeros((size, size), np.int16)
    n = max(1, int(size * size * density))
    idx = rng.choice(size * size, n, replace=False)
    # energy concentrated at low frequencies like real quantised TUs, some large levels
    vals = rng.integers(1, 4 if not big else 40, n) * rng.choice([-1, 1], n)
    c.reshape(-1)[idx] = vals
    if rng.random() < 0.5:
        yy, xx = np.mgrid[0:size, 0:size]
        c[(yy + xx) > size * rng.uniform(0.3, 1.5)] = 0
    if not c.any():
        c[0, 0] = 3
    return c


def decl(oracle):
    oracle.svt_oracle_coeff_bits_lossy.restype = u64
    oracle.svt_oracle_coeff_bits_lossy.argtypes = [vp, u32, u32, u32, u32, vp, u32, u32, u32]


def test_cost_layout():
    assert COST.itemsize == 1560


@pytest.mark.parametrize("size", [4, 8, 16, 32])
def test_rate_matches_reference(oracle, size):
    decl(oracle)
    rng = np.random.default_rng(size)
    ref.EstimateQuantizedCoefficients_Lossy.restype = C.c_int
    checked = 0
    for trial in range(300):
        cost = make_cost(trial % 7)
        tu = random_tu(rng, size, rng.choice([0.02, 0.1, 0.4, 0.9]), trial % 3 == 0)
        if trial % 10 == 0:  # DC-only fast track
            tu[:] = 0
            tu[0, 0] = rng.integers(1, 9) * rng.choice([-1, 1])
        buf = np.zeros((size, 40), np.int16)
        buf[:, :size] = tu
        nnz = int(np.count_nonzero(tu))
        typ = 2 if trial % 2 else 1
        luma_mode, chroma_mode, comp = int(rng.integers(0, 35)), int(rng.integers(0, 5)), int(rng.integers(0, 3))
        if comp and size == 32:
            comp = 0  # chroma TUs are at most 16x16 in the tables' offset scheme... keep what the encoder can call
        want = u64(12345)
        rc = ref.EstimateQuantizedCoefficients_Lossy(vp(cost.ctypes.data), None, u32(size), u32(typ), u32(luma_mode), u32(chroma_mode),
                                                     vp(buf.ctypes.data), u32(40), u32(comp), u32(nnz), C.byref(want))
        assert rc == 0
        got = oracle.svt_oracle_coeff_bits_lossy(cost.ctypes.data, size, typ, luma_mode, chroma_mode, buf.ctypes.data, 40, comp, nnz)
        assert got == want.value - 12345, (size, trial, typ, luma_mode, comp, nnz)
        checked += 1
    assert checked == 300


# ---- the CABAC-context-updating estimator (coeffCabacUpdate) ------------------------------------------------------------
CTX_WORDS = 136  # CoeffCtxtMdl_t: lastSigX[30] lastSigY[30] sig[42] coeffGroupSig[4] greater1[24] greater2[6]


def decl_update(oracle):
    oracle.svt_oracle_coeff_bits_update.restype = u64
    oracle.svt_oracle_coeff_bits_update.argtypes = [vp, u32, u32, u32, u32, vp, u32, u32, u32]


def test_update_tables_are_the_reference_tables(oracle):
    """The generated state-transition table (H.265 Table 9-41) and the embedded entropy-bit constants equal the arrays the
    reference exports (Codec/EbHmCode.c:216-246)."""
    decl_update(oracle)
    buf = np.zeros((4, 4), np.int16)
    buf[0, 0] = 1
    ctx = np.zeros(CTX_WORDS, np.uint32)
    oracle.svt_oracle_coeff_bits_update(ctx.ctypes.data, 4, 1, 0, 0, buf.ctypes.data, 4, 0, 1)  # builds the tables
    mine_next = np.ctypeslib.as_array((u32 * 256).in_dll(oracle, "svt_oracle_next_state_mps_lps"))
    mine_bits = np.ctypeslib.as_array((u32 * 128).in_dll(oracle, "svt_oracle_cabac_estimated_bits"))
    ref_next = np.ctypeslib.as_array((u32 * 256).in_dll(ref, "NextStateMpsLps"))
    ref_bits = np.ctypeslib.as_array((u32 * 128).in_dll(ref, "CabacEstimatedBits"))
    assert np.array_equal(mine_next, ref_next)
    assert np.array_equal(mine_bits, ref_bits)


@pytest.mark.parametrize("fn", ["EstimateQuantizedCoefficients_generic_Update", "EstimateQuantizedCoefficients_Update_SSE2"])
@pytest.mark.parametrize("size", [4, 8, 16, 32])
def test_update_rate_and_states_match_reference(oracle, size, fn):
    """Bits AND the updated context states, from random starting states, chained over several TUs like the mode decision
    chains them (the states one call leaves are the next call's input)."""
    decl_update(oracle)
    rng = np.random.default_rng(100 + size)
    f = getattr(ref, fn)
    f.restype = C.c_int
    for chain in range(60):
        ctx_ref = rng.integers(0, 126, CTX_WORDS).astype(np.uint32)
        ctx_mine = ctx_ref.copy()
        for trial in range(5):
            tu = random_tu(rng, size, rng.choice([0.02, 0.1, 0.4, 0.9]), (chain + trial) % 3 == 0)
            if (chain + trial) % 7 == 0:  # DC-only fast track
                tu[:] = 0
                tu[0, 0] = rng.integers(1, 9) * rng.choice([-1, 1])
            buf = np.zeros((size, 40), np.int16)
            buf[:, :size] = tu
            nnz = int(np.count_nonzero(tu))
            typ = 2 if (chain + trial) % 2 else 1
            luma_mode, chroma_mode, comp = int(rng.integers(0, 35)), int(rng.integers(0, 5)), int(rng.integers(0, 3))
            if comp and size == 32:
                comp = 0
            want = u64(777)
            rc = f(vp(ctx_ref.ctypes.data), None, None, u32(size), u32(typ), u32(luma_mode), u32(chroma_mode), vp(buf.ctypes.data),
                   u32(40), u32(comp), u32(nnz), C.byref(want))
            assert rc == 0
            got = oracle.svt_oracle_coeff_bits_update(ctx_mine.ctypes.data, size, typ, luma_mode, chroma_mode, buf.ctypes.data, 40, comp, nnz)
            assert got == want.value - 777, (fn, size, chain, trial, typ, luma_mode, comp, nnz)
            assert np.array_equal(ctx_mine, ctx_ref), (fn, size, chain, trial)
