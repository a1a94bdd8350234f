"""AC energy of the LCUs on the device (svt-hevc_amd/csrc/pa_kernels.hip k_sbo_ac_energy) through the C-ABI: svt_amd_picture_ac_energy of an
uploaded picture against the CPU checker (pinned on the reference's ComputeNxMSatdSadLCU by tests/test_oracle_sbo.py) and, where the box carries
oracle/_ref, against that reference symbol itself."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_sbo import oracle_picture

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def device_energy(lib, luma, w, h):
    lib.svt_amd_picture_ac_energy.restype, lib.svt_amd_picture_ac_energy.argtypes = C.c_int, [vp, C.c_int, vp]
    ctx = vp()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        luma = np.ascontiguousarray(luma)
        assert lib.svt_amd_picture_upload(ctx, 0, luma.ctypes.data, luma.shape[1], w, h) == 0, lib.svt_amd_last_error()
        out = np.zeros((S.lcu_count(w, h), 5), np.uint64)
        assert lib.svt_amd_picture_ac_energy(ctx, 0, out.ctypes.data) == 0, lib.svt_amd_last_error()
        return out
    finally:
        lib.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("kind,w,h", [("objects", 416, 240), ("noise", 640, 384), ("motion", 1920, 1080), ("objects", 3840, 2160), ("static", 832, 480)])
def test_device_energy_matches_the_checker(product, oracle, kind, w, h):
    luma = S.gen_luma(kind, w, h, 3, 21)
    out = device_energy(product, luma, w, h)
    want = oracle_picture(oracle, luma, w, h)
    assert np.array_equal(out, want), np.argwhere(out != want)[:4].tolist()
    assert (out[:, 0] < 100000000).sum() == (w // 64) * (h // 64)


def test_device_energy_of_extreme_pictures(product, oracle):
    rng = np.random.default_rng(4)
    w, h = 256, 128
    for luma in (np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8), np.tile(np.array([[0, 255], [255, 0]], np.uint8), (h // 2, w // 2)),
                 rng.integers(0, 256, (h, w), dtype=np.uint8)):
        assert np.array_equal(device_energy(product, luma, w, h), oracle_picture(oracle, luma, w, h))


def test_device_energy_matches_the_reference_symbol(product):
    ref = S.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libsvtref.so not on this box")
    ref.ComputeNxMSatdSadLCU.restype, ref.ComputeNxMSatdSadLCU.argtypes = C.c_uint64, [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    w, h = 640, 384
    luma = np.ascontiguousarray(S.gen_luma("objects", w, h, 7, 2))
    out = device_energy(product, luma, w, h)
    for k in range(out.shape[0]):
        x, y = 64 * (k % 10), 64 * (k // 10)
        assert out[k, 0] == ref.ComputeNxMSatdSadLCU(luma[y:, x:].ctypes.data, w, 64, 64)
        for q in range(4):
            assert out[k, 1 + q] == ref.ComputeNxMSatdSadLCU(luma[y + 32 * (q >> 1):, x + 32 * (q & 1):].ctypes.data, w, 32, 32)
