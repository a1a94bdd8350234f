"""-m gpu: coefficient rate estimation through the C-ABI (leaf entry point + batched form) vs the oracle (pinned to the
reference's EstimateQuantizedCoefficients_Lossy in tests/test_oracle_rate.py)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_rate import COST, decl, random_tu

pytestmark = pytest.mark.gpu
u32, vp, u64 = C.c_uint32, C.c_void_p, C.c_uint64
TU_INFO = np.dtype([("num_nonzero", "<u4"), ("type", "u1"), ("luma", "u1"), ("chroma", "u1"), ("component", "u1")])


def synthetic_cost(seed):
    """Cost tables with the structure PrecomputeCabacCost produces, without needing the reference on the GPU box:
    7-bit context states -> per-bin bit estimates that differ for bin 0 / bin 1."""
    rng = np.random.default_rng(seed)
    cost = np.zeros(1, COST)
    cost["last"] = rng.integers(0, 4000, 176)
    for name in ("sig", "g1", "g2", "sigml"):
        cost[name] = rng.integers(2, 200, cost[name].shape)
    cost["g1x"] = np.cumsum(rng.integers(2, 60, (6, 16)), axis=1).reshape(-1)
    cost["sigv"] = rng.integers(0, 255, (32, 16))
    return cost


def make_batch(rng, size, n):
    tus = np.zeros((n, size, size), np.int16)
    info = np.zeros(n, TU_INFO)
    for i in range(n):
        tu = random_tu(rng, size, rng.choice([0.02, 0.1, 0.4, 0.9]), i % 3 == 0)
        if i % 10 == 0:
            tu[:] = 0
            tu[0, 0] = rng.integers(1, 9) * rng.choice([-1, 1])
        if i % 17 == 5:
            tu[:] = 0  # all-zero TU: 0 bits by definition of the batched form
        tus[i] = tu
        comp = int(rng.integers(0, 3)) if size < 32 else 0
        info[i] = (np.count_nonzero(tu), 2 if i % 2 else 1, rng.integers(0, 35), rng.integers(0, 5), comp)
    return tus, info


@pytest.mark.parametrize("size", [4, 8, 16, 32])
def test_rate_batch_matches_oracle(product, gpu_ctx, oracle, size):
    import torch
    decl(oracle)
    product.svt_amd_coeff_bits_batch.argtypes = [vp, vp, u32, vp, vp, vp, u32]
    rng = np.random.default_rng(size)
    n = 1500
    tus, info = make_batch(rng, size, n)
    for seed in (1, 2):
        cost = synthetic_cost(seed)
        d_c, d_i = torch.from_numpy(tus).cuda(), torch.from_numpy(info.view(np.uint8).copy()).cuda()
        d_o = torch.zeros(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        rc = product.svt_amd_coeff_bits_batch(gpu_ctx, cost.ctypes.data, size, d_c.data_ptr(), d_i.data_ptr(), d_o.data_ptr(), n)
        assert rc == 0, product.svt_amd_last_error()
        product.svt_amd_synchronize(gpu_ctx)
        got = d_o.cpu().numpy().astype(np.uint64)
        for i in range(n):
            nz = int(info[i]["num_nonzero"])
            want = 0 if nz == 0 else oracle.svt_oracle_coeff_bits_lossy(
                cost.ctypes.data, size, int(info[i]["type"]), int(info[i]["luma"]), int(info[i]["chroma"]),
                tus[i].ctypes.data, size, int(info[i]["component"]), nz)
            assert int(got[i]) == want, (size, seed, i, info[i])


def test_rate_leaf(product, oracle):
    decl(oracle)
    product.svt_amd_EstimateQuantizedCoefficients_Lossy.restype = C.c_int
    rng = np.random.default_rng(9)
    cost = synthetic_cost(3)
    for size in (4, 8, 16, 32):
        for trial in range(12):
            tu = random_tu(rng, size, rng.choice([0.05, 0.5]), trial % 2 == 0)
            buf = np.zeros((size, 40), np.int16)
            buf[:, :size] = tu
            nnz = int(np.count_nonzero(tu))
            typ, lm, cm = 2 if trial % 2 else 1, int(rng.integers(0, 35)), int(rng.integers(0, 5))
            acc = u64(777)
            rc = product.svt_amd_EstimateQuantizedCoefficients_Lossy(vp(cost.ctypes.data), None, u32(size), u32(typ), u32(lm), u32(cm),
                                                                     vp(buf.ctypes.data), u32(40), u32(0), u32(nnz), C.byref(acc))
            assert rc == 0, product.svt_amd_last_error()
            want = oracle.svt_oracle_coeff_bits_lossy(cost.ctypes.data, size, typ, lm, cm, buf.ctypes.data, 40, 0, nnz)
            assert acc.value - 777 == want, (size, trial)


# ---- the CABAC-context-updating estimator (coeffCabacUpdate) ------------------------------------------------------------
from test_oracle_rate import CTX_WORDS, decl_update  # noqa: E402


@pytest.mark.parametrize("size,chain", [(4, 4), (8, 3), (16, 4), (32, 2), (32, 4)])
def test_update_rate_batch_matches_oracle(product, gpu_ctx, oracle, size, chain):
    """Bits of every block and the final states of every chain (units of one candidate share a model and are walked in order)."""
    import torch
    decl_update(oracle)
    product.svt_amd_coeff_bits_update_batch.restype = C.c_int
    product.svt_amd_coeff_bits_update_batch.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32]
    rng = np.random.default_rng(40 + size + chain)
    n = 602  # last chain is short
    tus, info = make_batch(rng, size, n)
    nch = (n + chain - 1) // chain
    ctx0 = rng.integers(0, 126, (nch, CTX_WORDS)).astype(np.uint32)
    d_c, d_i = torch.from_numpy(tus).cuda(), torch.from_numpy(info.view(np.uint8).copy()).cuda()
    d_m = torch.from_numpy(ctx0.view(np.int32).copy()).cuda()
    d_o = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rc = product.svt_amd_coeff_bits_update_batch(gpu_ctx, size, d_c.data_ptr(), d_i.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), n, chain)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    got = d_o.cpu().numpy().astype(np.uint64)
    got_m = d_m.cpu().numpy().view(np.uint32)
    for c in range(nch):
        m = ctx0[c].copy()
        for i in range(c * chain, min(n, (c + 1) * chain)):
            nz = int(info[i]["num_nonzero"])
            want = 0 if nz == 0 else oracle.svt_oracle_coeff_bits_update(
                m.ctypes.data, size, int(info[i]["type"]), int(info[i]["luma"]), int(info[i]["chroma"]), tus[i].ctypes.data, size,
                int(info[i]["component"]), nz)
            assert int(got[i]) == want, (size, chain, i, info[i])
        assert np.array_equal(got_m[c], m), (size, chain, c)


def test_update_rate_leaf(product, oracle):
    decl_update(oracle)
    product.svt_amd_EstimateQuantizedCoefficients_Update.restype = C.c_int
    rng = np.random.default_rng(19)
    for size in (4, 8, 16, 32):
        m_dev = rng.integers(0, 126, CTX_WORDS).astype(np.uint32)
        m_ora = m_dev.copy()
        for trial in range(6):
            tu = random_tu(rng, size, rng.choice([0.05, 0.5]), trial % 2 == 0)
            buf = np.zeros((size, 40), np.int16)
            buf[:, :size] = tu
            nnz = int(np.count_nonzero(tu))
            typ, lm, cm = 2 if trial % 2 else 1, int(rng.integers(0, 35)), int(rng.integers(0, 5))
            comp = int(rng.integers(0, 3)) if size < 32 else 0
            acc = u64(555)
            rc = product.svt_amd_EstimateQuantizedCoefficients_Update(vp(m_dev.ctypes.data), None, None, u32(size), u32(typ), u32(lm), u32(cm),
                                                                      vp(buf.ctypes.data), u32(40), u32(comp), u32(nnz), C.byref(acc))
            assert rc == 0, product.svt_amd_last_error()
            want = oracle.svt_oracle_coeff_bits_update(m_ora.ctypes.data, size, typ, lm, cm, buf.ctypes.data, 40, comp, nnz)
            assert acc.value - 555 == want, (size, trial)
            assert np.array_equal(m_dev, m_ora), (size, trial)
