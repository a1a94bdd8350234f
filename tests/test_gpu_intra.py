"""-m gpu: intra-prediction kernels through the C-ABI (24 leaf entry points + batched form) vs the oracle
(pinned to the reference's C_DEFAULT kernels in tests/test_oracle_intra.py)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_intra import KERNELS

pytestmark = pytest.mark.gpu
u32, i32, vp = C.c_uint32, C.c_int32, C.c_void_p


@pytest.fixture(scope="module")
def libs(product, oracle):
    oracle.svt_oracle_IntraPred.argtypes = [C.c_int, C.c_int, u32, vp, vp, u32, C.c_int, i32]
    product.svt_amd_intra_pred_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, i32, vp, u32, i32, vp, u32]
    return product, oracle


def refs_for(size, bps, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256 if bps == 1 else 1024, size=8 * size + 16).astype(np.uint8 if bps == 1 else np.uint16)


@pytest.mark.parametrize("mode", sorted(KERNELS))
@pytest.mark.parametrize("bps", [1, 2])
def test_intra_leaf(libs, mode, bps):
    product, oracle = libs
    name, ang = KERNELS[mode][bps - 1], KERNELS[mode][2]
    fn = getattr(product, "svt_amd_" + name)
    dt = np.uint8 if bps == 1 else np.uint16
    for size in (4, 8, 16, 32, 64):
        for skip in (0, 1):
            for angle in ([32, 13, 2, -9, -32] if ang else [0]):
                refs = refs_for(size, bps, size + mode)
                base = refs.ctypes.data + ((2 * size + 4) * bps if ang else 0)
                want = np.full((size, size + 8), 77, dt)
                got = want.copy()
                oracle.svt_oracle_IntraPred(mode, bps, size, base, want.ctypes.data, size + 8, skip, angle)
                if ang:
                    fn(u32(size), vp(base), vp(got.ctypes.data), u32(size + 8), C.c_ubyte(skip), i32(angle))
                else:
                    fn(u32(size), vp(base), vp(got.ctypes.data), u32(size + 8), C.c_ubyte(skip))
                assert np.array_equal(got, want), (name, size, skip, angle)


@pytest.mark.parametrize("bps", [1, 2])
def test_intra_batched(libs, gpu_ctx, bps):
    """300 blocks per launch, every kernel, device pointers; each block vs the oracle."""
    import torch
    product, oracle = libs
    dev = torch.device("cuda", 0)
    n, size = 300, 16
    pitch = 8 * size + 16
    rng = np.random.default_rng(bps)
    dt = np.uint8 if bps == 1 else np.uint16
    refs = rng.integers(0, 256 if bps == 1 else 1024, size=(n, pitch)).astype(dt)
    d_refs = torch.from_numpy(refs.view(np.int16) if bps == 2 else refs).to(dev)
    d_pred = torch.zeros((n, size, size), dtype=torch.int16 if bps == 2 else torch.uint8, device=dev)
    torch.cuda.synchronize()
    for mode in sorted(KERNELS):
        ang = KERNELS[mode][2]
        angle = -17 if ang else 0
        off = 2 * size + 4 if ang else 0
        torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
        assert product.svt_amd_intra_pred_batch(gpu_ctx, mode, bps, size, 0, angle, d_refs.data_ptr(), pitch, off,
                                                d_pred.data_ptr(), n) == 0
        assert product.svt_amd_synchronize(gpu_ctx) == 0
        got = d_pred.cpu().numpy().view(dt)
        for b in range(0, n, 7):
            want = np.zeros((size, size), dt)
            oracle.svt_oracle_IntraPred(mode, bps, size, refs[b].ctypes.data + off * bps, want.ctypes.data, size, 0, angle)
            assert np.array_equal(got[b], want), (mode, b)
