"""-m gpu: the LEAF layer of the C-ABI (table-slot replacements) against the oracle."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

pytestmark = pytest.mark.gpu
u32, u64, i16 = C.c_uint32, C.c_uint64, C.c_int16


def P(a):
    return a.ctypes.data


def rnd(rng, *shape):
    return rng.integers(0, 256, size=shape, dtype=np.uint8)


@pytest.mark.parametrize("w,h", [(8, 8), (16, 16), (64, 32), (24, 12), (64, 64), (10, 8)])
def test_sad_family(product, oracle, w, h):
    rng = np.random.default_rng(w + h)
    a, b, c = rnd(rng, 80, 96), rnd(rng, 80, 128), rnd(rng, 80, 72)
    assert product.svt_amd_NxMSadKernel(P(a), 96, P(b), 128, h, w) == \
        oracle.svt_oracle_NxMSadKernel(P(a), 96, P(b), 128, h, w)
    assert product.svt_amd_NxMSadAveragingKernel(P(a), 96, P(b), 128, P(c), 72, h, w) == \
        oracle.svt_oracle_NxMSadAveragingKernel(P(a), 96, P(b), 128, P(c), 72, h, w)
    assert product.svt_amd_SpatialFullDistortionKernel(P(a), 96, P(b), 128, w, h) == \
        oracle.svt_oracle_SpatialFullDistortionKernel(P(a), 96, P(b), 128, w, h)


@pytest.mark.parametrize("kind", ["random", "flat"])
@pytest.mark.parametrize("w,h,saw,sah", [(16, 8, 24, 20), (32, 16, 8, 4), (64, 32, 16, 9), (10, 8, 7, 3)])
def test_sad_loop(product, oracle, kind, w, h, saw, sah):
    rng = np.random.default_rng(5)
    src, refp = rnd(rng, 64, 64), rnd(rng, 160, 256)
    if kind == "flat":
        src[:] = 128
        refp[:] = 128
    out = []
    for fn in (oracle.svt_oracle_SadLoopKernel, product.svt_amd_SadLoopKernel):
        best, x, y = u64(0), i16(-7), i16(-9)
        fn(P(src), 128, P(refp), 512, h, w, C.byref(best), C.byref(x), C.byref(y), 256, saw, sah)
        out.append((best.value, x.value, y.value))
    assert out[0] == out[1]


@pytest.mark.parametrize("seed", range(3))
def test_search_point_kernels(product, oracle, seed):
    rng = np.random.default_rng(seed)
    src, refp = rnd(rng, 16, 64), rnd(rng, 16, 96)
    all16 = rng.integers(0, 32641, size=128).astype(np.uint16)
    all32 = rng.integers(0, 65000, size=16).astype(np.uint32)
    if seed == 2:
        src[:] = 9
        refp[:] = 9
        all16[:] = 50
    res = []
    for pre, lib in (("svt_oracle_", oracle), ("svt_amd_", product)):
        g = lambda n: getattr(lib, pre + n)
        bs8, bm8 = np.full(4, 2000, np.uint32), np.zeros(4, np.uint32)
        bs16, bm16 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        s16 = np.zeros(8, np.uint16)
        mv = ((-12 & 0xffff) << 16) | (20 & 0xffff)
        g("GetEightHorizontalSearchPointResults_8x8_16x16_PU")(P(src), 64, P(refp), 96, P(bs8), P(bm8), P(bs16),
                                                                 P(bm16), mv, P(s16))
        bs32, bm32 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        bs64, bm64 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        g("GetEightHorizontalSearchPointResults_32x32_64x64")(P(all16), P(bs32), P(bs64), P(bm32), P(bm64), mv)
        cs8, cm8 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        cs16, cm16, o16 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        g("SadCalculation_8x8_16x16")(P(src), 64, P(refp), 96, P(cs8), P(cs16), P(cm8), P(cm16), mv, P(o16))
        ds32, dm32 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        ds64, dm64 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        g("SadCalculation_32x32_64x64")(P(all32), P(ds32), P(ds64), P(dm32), P(dm64), mv)
        res.append([x.tolist() for x in (bs8, bm8, bs16, bm16, s16, bs32, bm32, bs64, bm64, cs8, cm8, cs16, cm16,
                                         o16, ds32, dm32, ds64, dm64)])
    assert res[0] == res[1]


def test_interpolation_average_decimate(product, oracle):
    rng = np.random.default_rng(3)
    img = rnd(rng, 48, 96)
    for frac in (1, 2, 3):
        for name in ("Horizontal", "Vertical"):
            outs = []
            for pre, lib in (("svt_oracle_", oracle), ("svt_amd_", product)):
                dst = np.zeros((32, 80), np.uint8)
                getattr(lib, pre + "AvcStyleLumaInterpolationFilter" + name)(P(img) + 4 * 96 + 4, 96, P(dst), 80, 72,
                                                                             30, None, frac)
                outs.append(dst)
            assert np.array_equal(outs[0], outs[1]), (frac, name)
    a, b = rnd(rng, 64, 64), rnd(rng, 64, 80)
    d0, d1 = np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8)
    oracle.svt_oracle_PictureAverageKernel(P(a), 64, P(b), 80, P(d0), 64, 64, 64)
    product.svt_amd_PictureAverageKernel(P(a), 64, P(b), 80, P(d1), 64, 64, 64)
    assert np.array_equal(d0, d1)
    big = rnd(rng, 64, 128)
    for step in (2, 4):
        o0, o1 = np.zeros((64 // step, 64), np.uint8), np.zeros((64 // step, 64), np.uint8)
        oracle.svt_oracle_Decimation2D(P(big), 128, 128, 64, P(o0), 64, step)
        product.svt_amd_Decimation2D(P(big), 128, 128, 64, P(o1), 64, step)
        assert np.array_equal(o0, o1)
