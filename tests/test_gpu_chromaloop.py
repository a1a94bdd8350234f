"""-m gpu: the fused chroma full-loop kernel through the C-ABI against (1) records of real FullLoop_R +
CuFullDistortionFastTuMode_R call pairs of the reference (tests/golden/chromaloop_*.npz) and (2) the oracle composite on
random candidates (pinned to the same records in tests/test_oracle_chromaloop_golden.py); batched and host-pointer
forms."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_chromaloop_golden import (CASES, ChromaLoopIn, ChromaLoopOut, check_out, load_chromaloop_case, record_in,
                                           record_planes)
from test_gpu_rate import synthetic_cost

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
IN_DT, OUT_DT = np.dtype(ChromaLoopIn), np.dtype(ChromaLoopOut)


def run_batch(product, gpu_ctx, cost, ins, residuals):
    """ins: list of ChromaLoopIn; residuals: list of (2, c, c) int16 arrays -> (outs, quants, recons)"""
    import torch
    n = len(ins)
    h_in = np.zeros(n, IN_DT)
    h_res = np.zeros((n, 2, 1024), np.int16)
    for i, (fin, r) in enumerate(zip(ins, residuals)):
        C.memmove(h_in[i:i + 1].ctypes.data, C.addressof(fin), C.sizeof(fin))
        h_res[i, :, :r[0].size] = r.reshape(2, -1)
    d_in = torch.from_numpy(h_in.view(np.uint8).copy()).cuda()
    d_res = torch.from_numpy(h_res).cuda()
    d_q, d_r = d_res.clone(), torch.zeros_like(d_res)   # the reference's quant buffer starts as the residual
    d_out = torch.zeros(n * OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
    product.svt_amd_full_loop_chroma_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
    torch.cuda.synchronize()
    rc = product.svt_amd_full_loop_chroma_batch(gpu_ctx, cost.ctypes.data, d_in.data_ptr(), d_res.data_ptr(), d_q.data_ptr(),
                                                d_r.data_ptr(), d_out.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    outs = d_out.cpu().numpy().view(OUT_DT)
    return outs, d_q.cpu().numpy(), d_r.cpu().numpy()


def as_struct(rec):
    o = ChromaLoopOut()
    C.memmove(C.addressof(o), rec.tobytes(), C.sizeof(o))
    return o


def planes_of(buf, c):
    return buf[:, :c * c].reshape(2, c, c)


def test_struct_sizes():
    assert C.sizeof(ChromaLoopIn) == 32 and C.sizeof(ChromaLoopOut) == 96


@pytest.mark.parametrize("name", CASES)
def test_chromaloop_matches_reference_golden(product, gpu_ctx, name):
    g = load_chromaloop_case(name)
    n = len(g["size"])
    keys = [g["cost"][i].tobytes() for i in range(n)]   # the CabacCost tables are per picture
    for key in sorted(set(keys)):
        idx = [i for i in range(n) if keys[i] == key]
        ins = [record_in(g, i) for i in idx]
        ress = [record_planes(g, i, "residual") for i in idx]
        cost = np.ascontiguousarray(g["cost"][idx[0]:idx[0] + 1])
        outs, qs, rs = run_batch(product, gpu_ctx, cost, ins, ress)
        for j, i in enumerate(idx):
            c = int(g["size"][i]) // 2
            check_out(g, i, as_struct(outs[j]), planes_of(qs[j], c), planes_of(rs[j], c), name)


def random_candidates(rng, count):
    ins, ress = [], []
    for k in range(count):
        size = int(rng.choice([8, 16, 32, 64]))
        fin = ChromaLoopIn()
        fin.size, fin.cb_qp, fin.cr_qp = size, int(rng.integers(8, 52)), int(rng.integers(8, 52))
        fin.slice_type, fin.pf_mode = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        fin.cand_type, fin.intra_luma_mode = int(rng.integers(1, 3)), int(rng.integers(0, 35))
        amp = int(rng.choice([1, 4, 30, 255]))
        c = size // 2
        res = rng.integers(-amp, amp + 1, (2, c, c)).astype(np.int16)
        if k % 5 == 0:  # smooth residual: energy in few coefficients
            res[0] = (np.add.outer(np.arange(c), np.arange(c)) * amp // c - amp // 2).astype(np.int16)
        ins.append(fin)
        ress.append(res)
    return ins, ress


def areas(fin):
    c = fin.size // 2
    T = 16 if fin.size == 64 else c
    return c, T, T >> (0 if T == 4 else (1 if (T == 8 and fin.pf_mode == 2) else fin.pf_mode))


def oracle_run(oracle, cost, fin, res):
    oracle.svt_oracle_full_loop_chroma.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_full_loop_chroma.restype = None
    res = np.ascontiguousarray(res)
    quant, recon, want = res.copy(), np.zeros_like(res), ChromaLoopOut()
    ptrs = [(C.c_void_p * 2)(a[0].ctypes.data, a[1].ctypes.data) for a in (res, quant, recon)]
    oracle.svt_oracle_full_loop_chroma(cost.ctypes.data, C.addressof(fin), ptrs[0], ptrs[1], ptrs[2], C.addressof(want))
    return quant, recon, want


def test_chromaloop_matches_oracle_random(product, gpu_ctx, oracle):
    """Random residuals over every size / slice type / PF mode (incl. N4) / candidate type / qp, incl. all-zero outcomes."""
    rng = np.random.default_rng(2)
    cost = synthetic_cost(6)
    ins, ress = random_candidates(rng, 600)
    outs, qs, rs = run_batch(product, gpu_ctx, cost, ins, ress)
    zero = nonzero = 0
    for k, (fin, res) in enumerate(zip(ins, ress)):
        quant, recon, want = oracle_run(oracle, cost, fin, res)
        got = as_struct(outs[k])
        c, T, ar = areas(fin)
        gq, gr = planes_of(qs[k], c), planes_of(rs[k], c)
        for p in range(2):
            for ty in range(0, c, T):
                for tx in range(0, c, T):
                    assert np.array_equal(gq[p][ty:ty + ar, tx:tx + ar], quant[p][ty:ty + ar, tx:tx + ar]), k
                    assert np.array_equal(gr[p][ty:ty + ar, tx:tx + ar], recon[p][ty:ty + ar, tx:tx + ar]), k
        assert bytes(got) == bytes(want), (k, fin.size, [list(v) for v in got.nz], [list(v) for v in want.nz],
                                           list(got.coeff_bits), list(want.coeff_bits))
        tot = sum(sum(v) for v in want.nz)
        zero += tot == 0
        nonzero += tot != 0
    assert zero > 20 and nonzero > 200


def test_chromaloop_host_pointer_form(product, gpu_ctx, oracle):
    """svt_amd_full_loop_chroma on pitch-32 host planes: only the quantised area is written back."""
    rng = np.random.default_rng(3)
    cost = synthetic_cost(7)
    ins, ress = random_candidates(rng, 40)
    product.svt_amd_full_loop_chroma.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp]
    for k, (fin, res) in enumerate(zip(ins, ress)):
        c, T, ar = areas(fin)
        quant, recon, want = oracle_run(oracle, cost, fin, res)
        hq = np.full((2, 32, 32), 12345, np.int16)
        hq[:, :c, :c] = res
        hr = np.full((2, 32, 32), -77, np.int16)
        before_q, before_r = hq.copy(), hr.copy()
        got = ChromaLoopOut()
        pq = (C.c_void_p * 2)(hq[0].ctypes.data, hq[1].ctypes.data)
        pr = (C.c_void_p * 2)(hr[0].ctypes.data, hr[1].ctypes.data)
        rc = product.svt_amd_full_loop_chroma(gpu_ctx, cost.ctypes.data, C.addressof(fin), pq, pq, pr, 32, C.addressof(got))
        assert rc == 0, product.svt_amd_last_error()
        assert bytes(got) == bytes(want), k
        mask = np.zeros((32, 32), bool)
        for ty in range(0, c, T):
            for tx in range(0, c, T):
                mask[ty:ty + ar, tx:tx + ar] = True
        for p in range(2):
            assert np.array_equal(hq[p][:c, :c][mask[:c, :c]], quant[p][mask[:c, :c]]), k
            assert np.array_equal(hr[p][:c, :c][mask[:c, :c]], recon[p][mask[:c, :c]]), k
            assert np.array_equal(hq[p][~mask], before_q[p][~mask]) and np.array_equal(hr[p][~mask], before_r[p][~mask]), k


def test_chromaloop_cabac_matches_oracle_random(product, gpu_ctx, oracle):
    """coeffCabacUpdate: Cb then Cr of every unit move the candidate's context model (TuEstimateCoeffBits_R,
    EbEntropyCoding.c:8032-8100); outputs and the updated models against the oracle composite."""
    import torch
    oracle.svt_oracle_full_loop_chroma_cabac.argtypes = [C.c_void_p] * 7
    oracle.svt_oracle_full_loop_chroma_cabac.restype = None
    rng = np.random.default_rng(12)
    cost = synthetic_cost(8)
    ins, ress = random_candidates(rng, 400)
    n = len(ins)
    models0 = rng.integers(0, 126, (n, 136)).astype(np.uint32)
    h_in = np.zeros(n, IN_DT)
    h_res = np.zeros((n, 2, 1024), np.int16)
    for i, (fin, r) in enumerate(zip(ins, ress)):
        C.memmove(h_in[i:i + 1].ctypes.data, C.addressof(fin), C.sizeof(fin))
        h_res[i, :, :r[0].size] = r.reshape(2, -1)
    d_in = torch.from_numpy(h_in.view(np.uint8).copy()).cuda()
    d_res = torch.from_numpy(h_res).cuda()
    d_q, d_r = d_res.clone(), torch.zeros_like(d_res)
    d_out = torch.zeros(n * OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
    d_m = torch.from_numpy(models0.view(np.int32).copy()).cuda()
    fn = product.svt_amd_full_loop_chroma_cabac_batch
    fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32]
    torch.cuda.synchronize()
    rc = fn(gpu_ctx, cost.ctypes.data, d_in.data_ptr(), d_res.data_ptr(), d_q.data_ptr(), d_r.data_ptr(), d_out.data_ptr(), d_m.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    outs = d_out.cpu().numpy().view(OUT_DT)
    models = d_m.cpu().numpy().view(np.uint32)
    for k, (fin, res) in enumerate(zip(ins, ress)):
        res = np.ascontiguousarray(res)
        quant, recon, want, m = res.copy(), np.zeros_like(res), ChromaLoopOut(), models0[k].copy()
        ptrs = [(C.c_void_p * 2)(a[0].ctypes.data, a[1].ctypes.data) for a in (res, quant, recon)]
        oracle.svt_oracle_full_loop_chroma_cabac(cost.ctypes.data, C.addressof(fin), ptrs[0], ptrs[1], ptrs[2], m.ctypes.data, C.addressof(want))
        got = as_struct(outs[k])
        assert bytes(got) == bytes(want), (k, fin.size, list(got.coeff_bits), list(want.coeff_bits))
        assert np.array_equal(models[k], m), (k, fin.size, "model")
