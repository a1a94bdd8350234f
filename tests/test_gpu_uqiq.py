"""-m gpu: the encode pass's quantiser (svt_amd_unified_quantize_batch / svt_amd_unified_quantize) through the C-ABI against
(1) records of real UnifiedQuantizeInvQuantize calls of the reference (tests/golden/uqiq_*.npz) and (2) the oracle (pinned to
those records and to the reference function in tests/test_oracle_uqiq_golden.py) on random units with every optional
branch."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_uqiq_golden import CASES, UNIT, blocks_of, load_uqiq_case, oracle_call, random_coeff, random_units, unit_of

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def run_units(product, gpu_ctx, units, coeffs, quant_in, recon_in):
    import torch
    n = len(units)
    slab = lambda blocks: np.stack([np.pad(b.reshape(-1), (0, 1024 - b.size)) for b in blocks]).astype(np.int16)
    d_u = torch.from_numpy(units.view(np.uint8).copy()).cuda()
    d_c, d_q, d_r = (torch.from_numpy(slab(x)).cuda() for x in (coeffs, quant_in, recon_in))
    d_nz = torch.zeros(n, dtype=torch.int32, device="cuda")
    product.svt_amd_unified_quantize_batch.argtypes = [vp, vp, vp, vp, vp, vp, u32]
    torch.cuda.synchronize()
    rc = product.svt_amd_unified_quantize_batch(gpu_ctx, d_u.data_ptr(), d_c.data_ptr(), d_q.data_ptr(), d_r.data_ptr(), d_nz.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    return d_q.cpu().numpy(), d_r.cpu().numpy(), d_nz.cpu().numpy()


@pytest.mark.parametrize("name", CASES)
def test_uqiq_matches_reference_golden(product, gpu_ctx, name):
    g = load_uqiq_case(name)
    n = len(g["size"])
    units = np.concatenate([unit_of(g, i) for i in range(n)])
    blk = [blocks_of(g, i) for i in range(n)]
    q, r, nz = run_units(product, gpu_ctx, units, [b[0] for b in blk], [b[1] for b in blk], [b[2] for b in blk])
    for i in range(n):
        s = int(g["size"][i])
        assert int(nz[i]) == int(g["nz_out"][i]), (name, i)
        assert np.array_equal(q[i, :s * s].reshape(s, s), blk[i][3]) and np.array_equal(r[i, :s * s].reshape(s, s), blk[i][4]), (name, i)


def test_uqiq_matches_oracle_random(product, gpu_ctx, oracle):
    rng = np.random.default_rng(21)
    units = random_units(rng, 1200)
    coeffs = [random_coeff(rng, int(units["size"][k]), k) for k in range(len(units))]
    qin = [rng.integers(-9, 9, c.shape).astype(np.int16) for c in coeffs]
    rin = [rng.integers(-9, 9, c.shape).astype(np.int16) for c in coeffs]
    q, r, nz = run_units(product, gpu_ctx, units, coeffs, qin, rin)
    cleaned = forced = 0
    for k in range(len(units)):
        s = int(units["size"][k])
        wq, wr = qin[k].copy(), rin[k].copy()
        wnz = oracle_call(oracle, units[k:k + 1], coeffs[k], wq, wr)
        assert int(nz[k]) == wnz, (k, units[k])
        assert np.array_equal(q[k, :s * s].reshape(s, s), wq) and np.array_equal(r[k, :s * s].reshape(s, s), wr), (k, units[k])
        forced += int(wnz == 1 and units["enable_cb_flag"][k] == 1)
    assert forced > 3


def test_uqiq_host_pointer_form(product, gpu_ctx, oracle):
    product.svt_amd_unified_quantize.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    rng = np.random.default_rng(22)
    units = random_units(rng, 40)
    for k in range(len(units)):
        s = int(units["size"][k])
        coeff = np.zeros((s, 64), np.int16)
        coeff[:, :s] = random_coeff(rng, s, k)
        gq, gr = np.full((s, 64), 77, np.int16), np.full((s, 64), -5, np.int16)
        wq, wr = gq.copy(), gr.copy()
        wnz = oracle_call(oracle, units[k:k + 1], coeff, wq, wr)
        gnz = C.c_uint32(0)
        rc = product.svt_amd_unified_quantize(gpu_ctx, units[k:k + 1].ctypes.data, coeff.ctypes.data, 64, gq.ctypes.data, gr.ctypes.data,
                                              C.byref(gnz))
        assert rc == 0, product.svt_amd_last_error()
        assert gnz.value == wnz and np.array_equal(gq, wq) and np.array_equal(gr, wr), (k, units[k])
