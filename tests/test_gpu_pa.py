"""Picture-analysis statistics on the device (svt-hevc_amd/csrc/pa_kernels.hip) through the C-ABI: svt_amd_picture_stats of an uploaded picture
against what the REFERENCE encoder gathered for the same picture (tests/golden/pa_*.npz) and, at BASELINE configs[2]'s size, against the CPU checker."""
import ctypes as C
import os

import numpy as np
import pytest

import svtlib as S
from test_oracle_pa import CASES, oracle_picture, padded_luma

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def device_stats(lib, ctx, luma, w, h):
    lib.svt_amd_picture_stats.restype = C.c_int
    lib.svt_amd_picture_stats.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]
    luma = np.ascontiguousarray(luma)
    assert lib.svt_amd_picture_upload(ctx, 0, luma.ctypes.data, luma.shape[1], w, h) == 0, lib.svt_amd_last_error()
    out = np.zeros(S.lcu_count(w, h), S.PA_LCU_STATS_DTYPE)
    hist, ravg, total = np.zeros((4, 4, 256), np.uint32), np.zeros((4, 4), np.uint8), C.c_uint64(0)
    assert lib.svt_amd_picture_stats(ctx, 0, out.ctypes.data, 4, 4, hist.ctypes.data, ravg.ctypes.data, C.byref(total)) == 0, lib.svt_amd_last_error()
    return out, hist, ravg, int(total.value)


@pytest.mark.parametrize("name", CASES)
def test_device_statistics_are_the_encoders(product, name):
    lib = product
    g = np.load(os.path.join(S.GOLDEN_DIR, "pa_%s.npz" % name))
    kind, w, h, n, seed = g["clip"][0], int(g["clip"][1]), int(g["clip"][2]), int(g["clip"][3]), int(g["clip"][4])
    ctx = vp()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        for i, p in enumerate(g["picture_number"].tolist()):
            out, hist, ravg, total = device_stats(lib, ctx, S.gen_luma(kind, w, h, int(p), seed), w, h)
            assert np.array_equal(out["variance"], g["variance"][i]), (name, p, np.argwhere(out["variance"] != g["variance"][i])[:4].tolist())
            assert np.array_equal(out["y_mean"], g["y_mean"][i]), (name, p)
            assert np.array_equal(hist, g["histogram"][i]) and np.array_equal(ravg, g["region_average"][i]), (name, p)
            assert int(g["average_intensity"][i]) == (total + ((w * h) >> 1)) // (w * h)
    finally:
        lib.svt_amd_context_destroy(ctx)


def test_device_statistics_of_a_4k_picture_match_the_checker(product, oracle):
    lib = product
    w, h = 3840, 2160
    ctx = vp()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        luma = S.gen_luma("objects", w, h, 2, 9)
        out, hist, ravg, total = device_stats(lib, ctx, luma, w, h)
        want, whist, wavg, wtotal = oracle_picture(oracle, np.ascontiguousarray(np.pad(luma, ((0, 64), (0, 64)), mode="edge")), w, h)
        assert np.array_equal(out["variance"], want["variance"]) and np.array_equal(out["y_mean"], want["y_mean"])
        assert np.array_equal(hist, whist) and np.array_equal(ravg, wavg) and total == wtotal
        assert int(hist.sum()) == ((w // 4) * (h // 4) + 16 * 256) * 16
    finally:
        lib.svt_amd_context_destroy(ctx)
