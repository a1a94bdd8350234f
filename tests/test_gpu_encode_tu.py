"""-m gpu: the whole transform unit of the final encode pass in one kernel (svt_amd_encode_tu_batch) against the composition
of the oracle functions that are pinned piecewise to the reference (residual, EstimateTransform = svt_oracle_FwdTransform,
svt_oracle_unified_quantize, svt_oracle_recon_tu): quantised coefficients, non-zero counts and the reconstruction must
match for every size, both bit depths, all qps and slice types."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_uqiq_golden import UNIT as QUNIT

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
EUNIT = np.dtype([("src_off", "<i4"), ("rec_off", "<i4"), ("qp", "u1"), ("slice_type", "u1"), ("pad", "u1", 2), ("dz_offset", "<u4")])


@pytest.mark.parametrize("size,bps", [(4, 1), (8, 1), (16, 1), (32, 1), (8, 2), (16, 2), (32, 2)])
def test_encode_tu_matches_oracle_composition(product, gpu_ctx, oracle, size, bps):
    import torch
    oracle.svt_oracle_FwdTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    oracle.svt_oracle_FwdTransform.restype = None
    oracle.svt_oracle_unified_quantize.argtypes = [vp, vp, u32, vp, vp, vp]
    oracle.svt_oracle_unified_quantize.restype = None
    oracle.svt_oracle_recon_tu.argtypes = [C.c_int, u32, C.c_int, C.c_int, vp, vp, u32, vp, u32]
    oracle.svt_oracle_recon_tu.restype = None
    rng = np.random.default_rng(size * 3 + bps)
    W, H = 256 + 8, 128
    dt, maxv = (np.uint8, 255) if bps == 1 else (np.uint16, 1023)
    src = rng.integers(0, maxv + 1, (H, W)).astype(dt)
    # prediction: source plus a residual whose energy varies per unit (flat / small / large)
    pos = [(x, y) for y in range(0, H, size) for x in range(0, 256, size)]
    n = len(pos)
    pred = src.astype(np.int64)
    for k, (x, y) in enumerate(pos):
        amp = [0, 2, 12, 90][k % 4] * (4 if bps == 2 else 1)
        pred[y:y + size, x:x + size] += rng.integers(-amp, amp + 1, (size, size))
    pred = np.clip(pred, 0, maxv).astype(dt)
    units = np.zeros(n, EUNIT)
    for k, (x, y) in enumerate(pos):
        units[k] = (y * W + x, y * W + x, int(rng.integers(4, 52)), int(rng.integers(0, 3)), (0, 0), int(rng.integers(1, 20)) if k % 7 == 0 else 0)
    # --- device
    to_t = (lambda a: torch.from_numpy(a.view(np.int16)).cuda()) if bps == 2 else (lambda a: torch.from_numpy(a).cuda())
    d_src, d_rec = to_t(src), to_t(pred.copy())
    d_u = torch.from_numpy(units.view(np.uint8).copy()).cuda()
    d_q = torch.zeros(n * size * size, dtype=torch.int16, device="cuda")
    d_nz = torch.zeros(n, dtype=torch.int32, device="cuda")
    product.svt_amd_encode_tu_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, u32, vp, u32, vp, vp, u32]
    torch.cuda.synchronize()
    rc = product.svt_amd_encode_tu_batch(gpu_ctx, bps, size, d_u.data_ptr(), d_src.data_ptr(), W, d_rec.data_ptr(), W, d_q.data_ptr(),
                                         d_nz.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    got_rec = d_rec.cpu().numpy()
    got_rec = got_rec.view(np.uint16) if bps == 2 else got_rec
    got_q, got_nz = d_q.cpu().numpy().reshape(n, size, size), d_nz.cpu().numpy()
    # --- oracle composition
    zero = nonzero = 0
    for k, (x, y) in enumerate(pos):
        res = np.ascontiguousarray(src[y:y + size, x:x + size].astype(np.int16) - pred[y:y + size, x:x + size].astype(np.int16))
        coeff = np.zeros((size, size), np.int16)
        oracle.svt_oracle_FwdTransform(1 if size >= 16 else 0, size, res.ctypes.data, size, coeff.ctypes.data, size, None, 0 if bps == 1 else 2)
        qu = np.zeros(1, QUNIT)
        qu["size"], qu["qp"], qu["bit_depth"], qu["slice_type"], qu["dz_offset"] = size, units["qp"][k], 8 if bps == 1 else 10, units["slice_type"][k], units["dz_offset"][k]
        wq, wr = np.zeros((size, size), np.int16), np.zeros((size, size), np.int16)
        nz = C.c_uint32(0)
        oracle.svt_oracle_unified_quantize(qu.ctypes.data, coeff.ctypes.data, size, wq.ctypes.data, wr.ctypes.data, C.byref(nz))
        p = np.ascontiguousarray(pred[y:y + size, x:x + size])
        want = np.zeros_like(p)
        oracle.svt_oracle_recon_tu(bps, size, 0, 0, wr.ctypes.data, p.ctypes.data, size, want.ctypes.data, size)
        assert int(got_nz[k]) == nz.value, (k, size)
        assert np.array_equal(got_q[k], wq), (k, size, "quant")
        assert np.array_equal(got_rec[y:y + size, x:x + size], want), (k, size, "recon")
        zero += nz.value == 0
        nonzero += nz.value != 0
    assert zero > 0 and nonzero > n // 3
    assert np.array_equal(got_rec[:, 256:], pred[:, 256:])


def test_encode_plane_as_the_bench_runs_it(product, gpu_ctx, oracle):
    """bench.py's DCT stage: a whole luma plane as 16x16 units (8x8 for the rows a multiple of 16 leaves over), reconstruction in
    place, against oracle/svt_oracle_fullloop.c:svt_oracle_encode_plane (bench.py's cpu_baseline leg)."""
    import torch
    W, H = 416, 248
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (H, W), dtype=np.uint8)
    pred = np.clip(src.astype(np.int16) + rng.integers(-25, 26, (H, W)), 0, 255).astype(np.uint8)
    eudt = np.dtype([("src_off", "<i4"), ("rec_off", "<i4"), ("qp", "u1"), ("slice_type", "u1"), ("pad", "u1", 2), ("dz", "<u4")])
    H16 = H // 16 * 16
    want = pred.copy()
    oracle.svt_oracle_encode_plane.restype = C.c_uint64
    oracle.svt_oracle_encode_plane.argtypes = [vp, vp] + [u32] * 7
    nz_want = oracle.svt_oracle_encode_plane(src.ctypes.data, want.ctypes.data, W, W, 0, H16, 16, 32, 1)
    nz_want += oracle.svt_oracle_encode_plane(src.ctypes.data, want.ctypes.data, W, W, H16, H - H16, 8, 32, 1)
    d_src, d_rec = torch.from_numpy(src).cuda(), torch.from_numpy(pred.copy()).cuda()
    d_q = torch.zeros(H * W, dtype=torch.int16, device="cuda")
    product.svt_amd_encode_tu_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, u32, vp, u32, vp, vp, u32]
    total = 0
    for size, y0, y1 in ((16, 0, H16), (8, H16, H)):
        ys, xs = np.meshgrid(np.arange(y0, y1, size), np.arange(0, W, size), indexing="ij")
        u = np.zeros(ys.size, eudt)
        u["src_off"] = u["rec_off"] = (ys * W + xs).ravel()
        u["qp"], u["slice_type"] = 32, 1
        d_u = torch.from_numpy(u.view(np.uint8)).cuda()
        d_nz = torch.zeros(len(u), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        rc = product.svt_amd_encode_tu_batch(gpu_ctx, 1, size, d_u.data_ptr(), d_src.data_ptr(), W, d_rec.data_ptr(), W, d_q.data_ptr(),
                                             d_nz.data_ptr(), len(u))
        assert rc == 0, product.svt_amd_last_error()
        product.svt_amd_synchronize(gpu_ctx)
        total += int(d_nz.sum().item())
    assert np.array_equal(d_rec.cpu().numpy(), want) and total == nz_want and total > 1000
