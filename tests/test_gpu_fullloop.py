"""-m gpu: the fused luma full-loop kernel through the C-ABI against (1) records of real ProductFullLoop calls of the
reference (tests/golden/fullloop_*.npz) and (2) the oracle composite on random candidates (pinned to the same records
in tests/test_oracle_fullloop_golden.py)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_fullloop_golden import CASES, FullLoopIn, FullLoopOut, check_out, load_fullloop_case, record_in
from test_gpu_rate import synthetic_cost

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
IN_DT, OUT_DT = np.dtype(FullLoopIn), np.dtype(FullLoopOut)


def run_batch(product, gpu_ctx, cost, ins, residuals, models=None):
    """ins: list of FullLoopIn; residuals: list of size x size int16 arrays -> (outs, quants, recons).  models: (n, 136) uint32
    context models -> the coeffCabacUpdate entry point; the updated models are written back into the array."""
    import torch
    n = len(ins)
    h_in = np.zeros(n, IN_DT)
    h_res = np.zeros((n, 4096), np.int16)
    for i, (fin, r) in enumerate(zip(ins, residuals)):
        C.memmove(h_in[i:i + 1].ctypes.data, C.addressof(fin), C.sizeof(fin))
        h_res[i, :r.size] = r.reshape(-1)
    d_in = torch.from_numpy(h_in.view(np.uint8).copy()).cuda()
    d_res = torch.from_numpy(h_res).cuda()
    d_q, d_r = d_res.clone(), torch.zeros_like(d_res)   # the reference's quant buffer starts as the residual
    d_out = torch.zeros(n * OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    if models is not None:
        d_m = torch.from_numpy(models.view(np.int32).copy()).cuda()
        torch.cuda.synchronize()
        fn = product.svt_amd_full_loop_luma_cabac_batch
        fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32]
        rc = fn(gpu_ctx, cost.ctypes.data, d_in.data_ptr(), d_res.data_ptr(), d_q.data_ptr(), d_r.data_ptr(), d_out.data_ptr(),
                d_m.data_ptr(), n)
        assert rc == 0, product.svt_amd_last_error()
        product.svt_amd_synchronize(gpu_ctx)
        models[:] = d_m.cpu().numpy().view(np.uint32)
        return d_out.cpu().numpy().view(OUT_DT), d_q.cpu().numpy(), d_r.cpu().numpy()
    # the plain entry point serves the candidates without PM-core, the other one those with it: a mixed batch needs both
    for fn, wanted in ((product.svt_amd_full_loop_luma_batch, any((f.pf_mode >> 16) == 0 for f in ins)),
                       (product.svt_amd_full_loop_luma_pmcore_batch, any((f.pf_mode >> 16) != 0 for f in ins))):
        if not wanted:
            continue
        fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
        rc = fn(gpu_ctx, cost.ctypes.data, d_in.data_ptr(), d_res.data_ptr(), d_q.data_ptr(), d_r.data_ptr(), d_out.data_ptr(), n)
        assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    outs = d_out.cpu().numpy().view(OUT_DT)
    return outs, d_q.cpu().numpy(), d_r.cpu().numpy()


def as_struct(rec):
    o = FullLoopOut()
    C.memmove(C.addressof(o), rec.tobytes(), C.sizeof(o))
    return o


def test_struct_sizes():
    assert C.sizeof(FullLoopIn) == 72 and C.sizeof(FullLoopOut) == 64


@pytest.mark.parametrize("name", CASES)
def test_fullloop_matches_reference_golden(product, gpu_ctx, name):
    g = load_fullloop_case(name)
    n = len(g["size"])
    # the CabacCost tables are per picture: batch the records that share one
    cab = g["cabac_update"] if "cabac_update" in g else np.zeros(n, np.uint32)
    keys = [g["cost"][i].tobytes() + bytes([int(cab[i] != 0)]) for i in range(n)]
    for key in sorted(set(keys)):
        idx = [i for i in range(n) if keys[i] == key]
        ins, ress = [], []
        for i in idx:
            a, b = int(g["offsets"][i]), int(g["offsets"][i + 1])
            size = int(g["size"][i])
            ins.append(record_in(g, i))
            ress.append(np.ascontiguousarray(g["residual"][a:b]).reshape(size, size))
        cost = np.ascontiguousarray(g["cost"][idx[0]:idx[0] + 1])
        models = np.ascontiguousarray(g["ctx_in"][idx]).copy() if cab[idx[0]] else None   # coeffCabacUpdate records
        outs, qs, rs = run_batch(product, gpu_ctx, cost, ins, ress, models)
        if models is not None:
            assert np.array_equal(models, g["ctx_out"][idx]), (name, "context models")
        for j, i in enumerate(idx):
            size = int(g["size"][i])
            check_out(g, i, as_struct(outs[j]), qs[j, :size * size].reshape(size, size), rs[j, :size * size].reshape(size, size), name)


def test_fullloop_matches_oracle_random(product, gpu_ctx, oracle):
    """Random residuals over every size / slice type / PF mode / candidate type / qp, incl. all-zero outcomes."""
    oracle.svt_oracle_product_full_loop_luma.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_product_full_loop_luma.restype = None
    rng = np.random.default_rng(1)
    cost = synthetic_cost(5)
    ins, ress = [], []
    for k in range(600):
        size = int(rng.choice([8, 16, 32, 64]))
        fin = FullLoopIn()
        fin.size, fin.qp, fin.slice_type = size, int(rng.integers(10, 52)), int(rng.integers(0, 3))
        fin.pf_mode = int(rng.integers(0, 2)) if size >= 16 else 0
        if k % 3 == 0:      # PM-core (encMode 1..4): the upper half of the word is SvtAmdFullLoopIn.pm_core = EB_PMCORE
            fin.pf_mode |= 2 << 16
        fin.cand_type, fin.intra_luma_mode = int(rng.integers(1, 3)), int(rng.integers(0, 35))
        fin.full_lambda = int(rng.integers(1000, 4000000))
        for j, v in enumerate(rng.integers(1000, 90000, 4)):
            fin.cbf_bits[j] = int(v)
        fin.ycbf, fin.coeff_bits = int(rng.integers(0, 2)) << 7, int(rng.integers(0, 5000))
        fin.dist[0], fin.dist[1] = int(rng.integers(0, 9000)), int(rng.integers(0, 9000))
        amp = int(rng.choice([1, 4, 30, 255]))
        res = rng.integers(-amp, amp + 1, (size, size)).astype(np.int16)
        if k % 5 == 0:  # smooth residual: energy in few coefficients
            res = (np.add.outer(np.arange(size), np.arange(size)) * amp // size - amp // 2).astype(np.int16)
        ins.append(fin)
        ress.append(res)
    outs, qs, rs = run_batch(product, gpu_ctx, cost, ins, ress)
    zero = nonzero = 0
    for k, (fin, res) in enumerate(zip(ins, ress)):
        size = fin.size
        quant, recon, want = res.copy(), np.zeros_like(res), FullLoopOut()
        oracle.svt_oracle_product_full_loop_luma(cost.ctypes.data, C.addressof(fin), np.ascontiguousarray(res).ctypes.data,
                                                 quant.ctypes.data, recon.ctypes.data, C.addressof(want))
        got = as_struct(outs[k])
        T = 32 if size == 64 else size
        ar = T >> (fin.pf_mode & 0xffff)
        gq, gr = qs[k, :size * size].reshape(size, size), rs[k, :size * size].reshape(size, size)
        for ty in range(0, size, T):
            for tx in range(0, size, T):
                assert np.array_equal(gq[ty:ty + ar, tx:tx + ar], quant[ty:ty + ar, tx:tx + ar]), k
                assert np.array_equal(gr[ty:ty + ar, tx:tx + ar], recon[ty:ty + ar, tx:tx + ar]), k
        assert bytes(got) == bytes(want), (k, size, list(got.nz), list(want.nz), got.coeff_bits, want.coeff_bits)
        zero += sum(want.nz) == 0
        nonzero += sum(want.nz) != 0
    assert zero > 20 and nonzero > 200


def test_fullloop_cabac_matches_oracle_random(product, gpu_ctx, oracle):
    """coeffCabacUpdate: random residuals x random context models, plain and PM-core mixed, all sizes (64x64 chains four units
    through one model): outputs and updated models against the oracle composite (pinned on reference records)."""
    oracle.svt_oracle_product_full_loop_luma_cabac.argtypes = [C.c_void_p] * 7
    oracle.svt_oracle_product_full_loop_luma_cabac.restype = None
    rng = np.random.default_rng(7)
    cost = synthetic_cost(9)
    ins, ress = [], []
    for k in range(400):
        size = int(rng.choice([8, 16, 32, 64]))
        fin = FullLoopIn()
        fin.size, fin.qp, fin.slice_type = size, int(rng.integers(10, 52)), int(rng.integers(0, 3))
        fin.pf_mode = int(rng.integers(0, 2)) if size >= 16 else 0
        if k % 3 == 0:
            fin.pf_mode |= 2 << 16
        fin.cand_type, fin.intra_luma_mode = int(rng.integers(1, 3)), int(rng.integers(0, 35))
        fin.full_lambda = int(rng.integers(1000, 4000000))
        for j, v in enumerate(rng.integers(1000, 90000, 4)):
            fin.cbf_bits[j] = int(v)
        fin.ycbf, fin.coeff_bits = int(rng.integers(0, 2)) << 7, int(rng.integers(0, 5000))
        fin.dist[0], fin.dist[1] = int(rng.integers(0, 9000)), int(rng.integers(0, 9000))
        amp = int(rng.choice([1, 4, 30, 255]))
        res = rng.integers(-amp, amp + 1, (size, size)).astype(np.int16)
        if k % 5 == 0:
            res = (np.add.outer(np.arange(size), np.arange(size)) * amp // size - amp // 2).astype(np.int16)
        ins.append(fin)
        ress.append(res)
    models0 = rng.integers(0, 126, (len(ins), 136)).astype(np.uint32)
    models = models0.copy()
    outs, qs, rs = run_batch(product, gpu_ctx, cost, ins, ress, models)
    for k, (fin, res) in enumerate(zip(ins, ress)):
        size = fin.size
        quant, recon, want, m = res.copy(), np.zeros_like(res), FullLoopOut(), models0[k].copy()
        oracle.svt_oracle_product_full_loop_luma_cabac(cost.ctypes.data, C.addressof(fin), np.ascontiguousarray(res).ctypes.data,
                                                       quant.ctypes.data, recon.ctypes.data, m.ctypes.data, C.addressof(want))
        got = as_struct(outs[k])
        assert bytes(got) == bytes(want), (k, size, list(got.nz), list(want.nz), got.coeff_bits, want.coeff_bits)
        assert np.array_equal(models[k], m), (k, size, "model")
