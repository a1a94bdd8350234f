"""-m gpu: the encode pass's PM-core quantiser (svt_amd_pmcore_quantize_batch / svt_amd_pmcore_quantize) through the C-ABI against
(1) records of real UnifiedQuantizeInvQuantize calls of encMode 2 / 3 / 4 encodes (tests/golden/uqiqpm_*.npz) and (2) the oracle
(pinned to the same records in tests/test_oracle_uqiqpm_golden.py) on random units of every size, depth, qp and component."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_gpu_rate import synthetic_cost
from test_oracle_uqiqpm_golden import CASES, UNIT, blocks_of, load_uqiqpm_case, unit_of

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def run_batch(product, gpu_ctx, cost, units, coeffs):
    import torch
    n = len(units)
    h_c = np.zeros((n, 1024), np.int16)
    for i, c in enumerate(coeffs):
        h_c[i, :c.size] = c.reshape(-1)
    d_u = torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).copy()).cuda()
    d_c = torch.from_numpy(h_c).cuda()
    d_q, d_r = torch.zeros_like(d_c), torch.zeros_like(d_c)
    d_nz = torch.zeros(n, dtype=torch.int32, device="cuda")
    product.svt_amd_pmcore_quantize_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
    torch.cuda.synchronize()
    rc = product.svt_amd_pmcore_quantize_batch(gpu_ctx, cost.ctypes.data, d_u.data_ptr(), d_c.data_ptr(), d_q.data_ptr(), d_r.data_ptr(),
                                               d_nz.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    return d_q.cpu().numpy(), d_r.cpu().numpy(), d_nz.cpu().numpy()


@pytest.mark.parametrize("name", CASES)
def test_pmcore_quantize_matches_reference_golden(product, gpu_ctx, name):
    g = load_uqiqpm_case(name)
    for tbl in range(len(g["cost_tables"])):
        idx = [i for i in range(len(g["size"])) if int(g["cost_index"][i]) == tbl]
        units = np.concatenate([unit_of(g, i) for i in idx])
        blocks = [blocks_of(g, i) for i in idx]
        q, r, nz = run_batch(product, gpu_ctx, g["cost_tables"][tbl:tbl + 1], units, [b[0] for b in blocks])
        for k, i in enumerate(idx):
            n = int(g["size"][i])
            assert np.array_equal(q[k, :n * n].reshape(n, n), blocks[k][1]), (name, i, "quant")
            assert np.array_equal(r[k, :n * n].reshape(n, n), blocks[k][2]), (name, i, "recon")
            assert int(nz[k]) == int(g["nz_out"][i]), (name, i)


def test_pmcore_quantize_matches_oracle_random(product, gpu_ctx, oracle):
    oracle.svt_oracle_pmcore_quantize.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_pmcore_quantize.restype = None
    rng = np.random.default_rng(9)
    cost = synthetic_cost(3)
    n = 500
    units, coeffs = np.zeros(n, UNIT), []
    for k in range(n):
        size = int(rng.choice([4, 8, 16, 32]))
        units[k]["size"], units[k]["qp"], units[k]["bit_depth"] = size, int(rng.integers(4, 52)), int(rng.choice([8, 10]))
        units[k]["slice_type"], units[k]["component"], units[k]["cand_type"] = int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
        units[k]["lambda"] = int(rng.integers(1000, 6000000))
        amp = int(rng.choice([3, 40, 400, 6000]))
        c = rng.integers(-amp, amp + 1, (size, size))
        c = (c * (rng.random((size, size)) < rng.choice([0.05, 0.3, 1.0]))).astype(np.int16)
        if k % 7 == 0:
            c[0, 0] = int(rng.integers(-32768, 32768))
        coeffs.append(c)
    q, r, nz = run_batch(product, gpu_ctx, cost, units, coeffs)
    changed = 0
    for k in range(n):
        size = int(units[k]["size"])
        wq, wr, wnz = np.zeros_like(coeffs[k]), np.zeros_like(coeffs[k]), np.zeros(1, np.uint32)
        oracle.svt_oracle_pmcore_quantize(cost.ctypes.data, units[k:k + 1].ctypes.data, coeffs[k].ctypes.data, wq.ctypes.data, wr.ctypes.data,
                                          wnz.ctypes.data)
        assert np.array_equal(q[k, :size * size].reshape(size, size), wq), (k, units[k])
        assert np.array_equal(r[k, :size * size].reshape(size, size), wr) and int(nz[k]) == int(wnz[0]), (k, units[k])
        changed += int(wnz[0]) != int((np.abs(coeffs[k]) > 0).sum())
    assert changed > 100


def test_pmcore_quantize_host_form(product, gpu_ctx):
    g = load_uqiqpm_case(CASES[0])
    product.svt_amd_pmcore_quantize.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    for i in range(0, len(g["size"]), 11):
        coeff, wq, wr = blocks_of(g, i)
        n = int(g["size"][i])
        pitch = n + 8
        c, q, r = (np.full((n, pitch), 77, np.int16) for _ in range(3))
        c[:, :n] = coeff
        nz, u = np.zeros(1, np.uint32), unit_of(g, i)
        cost = g["cost_tables"][int(g["cost_index"][i]):int(g["cost_index"][i]) + 1]
        rc = product.svt_amd_pmcore_quantize(gpu_ctx, cost.ctypes.data, u.ctypes.data, c.ctypes.data, pitch, q.ctypes.data, r.ctypes.data, nz.ctypes.data)
        assert rc == 0, product.svt_amd_last_error()
        assert np.array_equal(q[:, :n], wq) and np.array_equal(r[:, :n], wr) and int(nz[0]) == int(g["nz_out"][i]), i
        assert (q[:, n:] == 77).all()
    bad = unit_of(g, 0)
    bad["size"] = 5
    assert product.svt_amd_pmcore_quantize(gpu_ctx, cost.ctypes.data, bad.ctypes.data, c.ctypes.data, pitch, q.ctypes.data, r.ctypes.data, nz.ctypes.data) != 0
