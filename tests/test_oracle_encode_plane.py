"""CPU-only: oracle/svt_oracle_fullloop.c:svt_oracle_encode_plane (bench.py's cpu_baseline leg for the DCT stage) is the plain
composition of functions pinned elsewhere (svt_oracle_FwdTransform, svt_oracle_unified_quantize, svt_oracle_recon_tu): checked here
unit by unit against that composition made from Python."""
import ctypes as C

import numpy as np

from test_oracle_uqiq_golden import UNIT as QUNIT

vp, u32 = C.c_void_p, C.c_uint32


def test_encode_plane_is_the_composition_of_pinned_functions(oracle):
    oracle.svt_oracle_encode_plane.restype = C.c_uint64
    oracle.svt_oracle_encode_plane.argtypes = [vp, vp] + [u32] * 7
    oracle.svt_oracle_FwdTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    oracle.svt_oracle_FwdTransform.restype = None
    oracle.svt_oracle_unified_quantize.argtypes = [vp, vp, u32, vp, vp, vp]
    oracle.svt_oracle_unified_quantize.restype = None
    oracle.svt_oracle_recon_tu.argtypes = [C.c_int, u32, C.c_int, C.c_int, vp, vp, u32, vp, u32]
    oracle.svt_oracle_recon_tu.restype = None
    rng = np.random.default_rng(2)
    W, H = 96, 56
    src = rng.integers(0, 256, (H, W), dtype=np.uint8)
    pred = np.clip(src.astype(np.int16) + rng.integers(-30, 31, (H, W)), 0, 255).astype(np.uint8)
    for size, row0, rows in ((16, 0, 48), (8, 48, 8), (32, 0, 32)):
        got = pred.copy()
        total = oracle.svt_oracle_encode_plane(src.ctypes.data, got.ctypes.data, W, W, row0, rows, size, 30, 1)
        want, nz_sum = pred.copy(), 0
        qu = np.zeros(1, QUNIT)
        qu["size"], qu["qp"], qu["bit_depth"], qu["slice_type"] = size, 30, 8, 1
        for y in range(row0, row0 + rows - size + 1, size):
            for x in range(0, W - size + 1, size):
                res = np.ascontiguousarray(src[y:y + size, x:x + size].astype(np.int16) - want[y:y + size, x:x + size].astype(np.int16))
                coeff, q, r, nz = np.zeros_like(res), np.zeros_like(res), np.zeros_like(res), C.c_uint32(0)
                oracle.svt_oracle_FwdTransform(1 if size >= 16 else 0, size, res.ctypes.data, size, coeff.ctypes.data, size, None, 0)
                oracle.svt_oracle_unified_quantize(qu.ctypes.data, coeff.ctypes.data, size, q.ctypes.data, r.ctypes.data, C.byref(nz))
                p = np.ascontiguousarray(want[y:y + size, x:x + size])
                out = np.zeros_like(p)
                oracle.svt_oracle_recon_tu(1, size, 0, 0, r.ctypes.data, p.ctypes.data, size, out.ctypes.data, size)
                want[y:y + size, x:x + size] = out
                nz_sum += nz.value
        assert np.array_equal(got, want) and total == nz_sum and total > 0, (size, total, nz_sum)
        assert np.array_equal(got[:row0], pred[:row0]) and np.array_equal(got[row0 + rows:], pred[row0 + rows:])
