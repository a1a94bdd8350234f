"""-m gpu: fused transform-unit reconstruction (svt_amd_recon_tu_batch) through the C-ABI against (1) records of real
EncodeGenerateRecon(16bit) calls of the reference (tests/golden/recon_*.npz) and (2) the oracle (pinned to the same records in
tests/test_oracle_recon_golden.py) on random units placed in a picture, in place and out of place."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_recon_golden import CASES, load_recon_case, record

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
UNIT = np.dtype([("pred_off", "<i4"), ("recon_off", "<i4"), ("only_dc", "u1"), ("dst", "u1"), ("pad", "u1", 2)])


def run_units(product, gpu_ctx, bps, size, coeffs, units, pred_plane, recon_plane=None):
    """coeffs (n, size, size) int16; units UNIT array; planes 2-D arrays -> reconstruction plane"""
    import torch
    to_t = (lambda a: torch.from_numpy(a.view(np.int16)).cuda()) if bps == 2 else (lambda a: torch.from_numpy(a).cuda())
    d_c = torch.from_numpy(np.ascontiguousarray(coeffs)).cuda()
    d_u = torch.from_numpy(units.view(np.uint8).copy()).cuda()
    d_p = to_t(np.ascontiguousarray(pred_plane))
    d_r = d_p if recon_plane is None else to_t(np.ascontiguousarray(recon_plane))
    product.svt_amd_recon_tu_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, u32, vp, u32, u32]
    torch.cuda.synchronize()
    rc = product.svt_amd_recon_tu_batch(gpu_ctx, bps, size, d_c.data_ptr(), d_u.data_ptr(), d_p.data_ptr(), pred_plane.shape[1],
                                        d_r.data_ptr(), (pred_plane if recon_plane is None else recon_plane).shape[1], len(units))
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    out = d_r.cpu().numpy()
    return out.view(np.uint16) if bps == 2 else out


@pytest.mark.parametrize("name", CASES)
def test_recon_matches_reference_golden(product, gpu_ctx, name):
    g = load_recon_case(name)
    recs = [record(g, i) for i in range(len(g["size"]))]
    for size in (4, 8, 16, 32):
        for bps in (1, 2):
            sel = [r for r in recs if r[0] == size and r[1] == bps]
            if not sel:
                continue
            dt = np.uint8 if bps == 1 else np.uint16
            # lay the units side by side in one plane, reconstruct in place like the reference
            plane = np.zeros((size, size * len(sel) + 3), dt)
            units = np.zeros(len(sel), UNIT)
            for k, r in enumerate(sel):
                plane[:, k * size:(k + 1) * size] = r[5]
                units[k] = (k * size, k * size, r[2], r[3], (0, 0))
            out = run_units(product, gpu_ctx, bps, size, np.stack([r[4] for r in sel]), units, plane)
            for k, r in enumerate(sel):
                assert np.array_equal(out[:, k * size:(k + 1) * size], r[6]), (name, size, bps, k, r[2], r[3])


@pytest.mark.parametrize("size,bps", [(4, 1), (8, 1), (16, 1), (32, 1), (4, 2), (16, 2), (32, 2)])
def test_recon_matches_oracle_random(product, gpu_ctx, oracle, size, bps):
    oracle.svt_oracle_recon_tu.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, vp, vp, u32, vp, u32]
    oracle.svt_oracle_recon_tu.restype = None
    rng = np.random.default_rng(size + bps)
    W, H = 256 + 5, 128
    dt, maxv = (np.uint8, 255) if bps == 1 else (np.uint16, 1023)
    pred = rng.integers(0, maxv + 1, (H, W)).astype(dt)
    pos = [(x, y) for y in range(0, H, size) for x in range(0, 256, size)]
    n = len(pos)
    amp = rng.choice([3, 40, 600, 32767], n)
    coeffs = np.stack([rng.integers(-a, a + 1, (size, size)) * (rng.random((size, size)) < 0.2) for a in amp]).astype(np.int16)
    units = np.zeros(n, UNIT)
    for k, (x, y) in enumerate(pos):
        units[k] = (y * W + x, y * W + x, int(rng.random() < 0.25 and size > 4), int(size == 4 and rng.random() < 0.5), (0, 0))
    want = np.zeros_like(pred)
    want[:, 256:] = 77
    for k, (x, y) in enumerate(pos):
        p = np.ascontiguousarray(pred[y:y + size, x:x + size])
        o = np.zeros_like(p)
        oracle.svt_oracle_recon_tu(bps, size, int(units[k]["only_dc"]), int(units[k]["dst"]), coeffs[k].ctypes.data, p.ctypes.data,
                                   size, o.ctypes.data, size)
        want[y:y + size, x:x + size] = o
    rec0 = np.zeros_like(pred)
    rec0[:, 256:] = 77
    got = run_units(product, gpu_ctx, bps, size, coeffs, units, pred, rec0)          # out of place
    assert np.array_equal(got, want)
    got2 = run_units(product, gpu_ctx, bps, size, coeffs, units, pred.copy())        # in place
    assert np.array_equal(got2[:, :256], want[:, :256]) and np.array_equal(got2[:, 256:], pred[:, 256:])
