"""CPU-only: pin the oracle's leaf kernels against the REFERENCE's own C_DEFAULT symbols
(oracle/_ref/libsvtref.so, built from /root/reference).  Skipped where that build is absent."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")

u32, u64, i16, vp = C.c_uint32, C.c_uint64, C.c_int16, C.c_void_p


def P(a):
    return a.ctypes.data


def rnd(rng, *shape):
    return rng.integers(0, 256, size=shape, dtype=np.uint8)


@pytest.mark.parametrize("w,h", [(8, 8), (16, 16), (64, 32), (24, 12), (40, 64), (64, 64)])
def test_nxm_sad(oracle, w, h):
    rng = np.random.default_rng(w * 100 + h)
    a, b = rnd(rng, 80, 96), rnd(rng, 80, 128)
    ref.FastLoop_NxMSadKernel.restype = u32
    want = ref.FastLoop_NxMSadKernel(vp(P(a)), 96, vp(P(b)), 128, h, w)
    assert oracle.svt_oracle_NxMSadKernel(P(a), 96, P(b), 128, h, w) == want


@pytest.mark.parametrize("kind", ["random", "flat", "ties"])
@pytest.mark.parametrize("w,h,saw,sah", [(16, 8, 24, 20), (32, 16, 8, 4), (64, 32, 16, 9), (10, 8, 7, 3)])
def test_sad_loop(oracle, kind, w, h, saw, sah):
    rng = np.random.default_rng(5)
    src = rnd(rng, 64, 64)
    refp = rnd(rng, 160, 256)
    if kind == "flat":
        src[:] = 128
        refp[:] = 128
    elif kind == "ties":
        refp[:] = np.tile(refp[:2, :4], (80, 64))
    out = []
    for fn in (ref.SadLoopKernel, oracle.svt_oracle_SadLoopKernel):
        best, x, y = u64(0), i16(-7), i16(-9)
        fn(vp(P(src)), u32(128), vp(P(refp)), u32(512), u32(h), u32(w), C.byref(best), C.byref(x), C.byref(y),
           u32(256), i16(saw), i16(sah))
        out.append((best.value, x.value, y.value))
    assert out[0] == out[1]


def test_sad_averaging(oracle):
    rng = np.random.default_rng(9)
    a, b, c = rnd(rng, 64, 64), rnd(rng, 70, 80), rnd(rng, 70, 96)
    ref.CombinedAveragingSAD.restype = u32
    for w, h in [(8, 8), (16, 8), (32, 32), (64, 64)]:
        want = ref.CombinedAveragingSAD(vp(P(a)), 64, vp(P(b)), 80, vp(P(c)), 96, h, w)
        assert oracle.svt_oracle_NxMSadAveragingKernel(P(a), 64, P(b), 80, P(c), 96, h, w) == want


@pytest.mark.parametrize("seed", range(4))
def test_eight_point_kernels(oracle, seed):
    rng = np.random.default_rng(seed)
    src, refp = rnd(rng, 16, 64), rnd(rng, 16, 96)
    if seed == 3:
        src[:] = 7
        refp[:] = 7
    res = []
    for fn8, fn32 in ((ref.GetEightHorizontalSearchPointResults_8x8_16x16_PU,
                       ref.GetEightHorizontalSearchPointResults_32x32_64x64),
                      (oracle.svt_oracle_GetEightHorizontalSearchPointResults_8x8_16x16_PU,
                       oracle.svt_oracle_GetEightHorizontalSearchPointResults_32x32_64x64)):
        bs8 = np.full(4, 3000 if seed != 3 else 0, np.uint32)
        bm8 = np.zeros(4, np.uint32)
        bs16 = np.full(1, 64 * 64 * 255, np.uint32)
        bm16 = np.zeros(1, np.uint32)
        s16 = np.zeros(8, np.uint16)
        mv = u32(((-12 & 0xffff) << 16) | (20 & 0xffff))
        fn8(vp(P(src)), u32(64), vp(P(refp)), u32(96), vp(P(bs8)), vp(P(bm8)), vp(P(bs16)), vp(P(bm16)), mv, vp(P(s16)))
        all16 = rng.integers(0, 32641, size=16 * 8).astype(np.uint16) if seed != 3 else np.zeros(128, np.uint16)
        bs32, bm32 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        bs64, bm64 = np.full(1, 0 if seed == 3 else 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        fn32(vp(P(all16)), vp(P(bs32)), vp(P(bs64)), vp(P(bm32)), vp(P(bm64)), mv)
        res.append([x.tolist() for x in (bs8, bm8, bs16, bm16, s16, bs32, bm32, bs64, bm64)])
        rng = np.random.default_rng(seed + 100)  # same all16 for both implementations
    # all16 differs between the two passes unless re-seeded identically: recompute deterministically
    assert res[0][:5] == res[1][:5]


@pytest.mark.parametrize("seed", range(3))
def test_tree_32_64_same_input(oracle, seed):
    rng = np.random.default_rng(seed)
    all16 = rng.integers(0, 32641, size=16 * 8).astype(np.uint16)
    if seed == 2:
        all16[:] = 100  # ties: 64x64 takes the LAST position ('<='), 32x32 the first
    res = []
    for fn in (ref.GetEightHorizontalSearchPointResults_32x32_64x64,
               oracle.svt_oracle_GetEightHorizontalSearchPointResults_32x32_64x64):
        bs32, bm32 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        bs64, bm64 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        fn(vp(P(all16)), vp(P(bs32)), vp(P(bs64)), vp(P(bm32)), vp(P(bm64)), u32(0x00080004))
        res.append([x.tolist() for x in (bs32, bm32, bs64, bm64)])
    assert res[0] == res[1]
    if seed == 2:
        assert res[0][1][0] == 0x00080004 and res[0][3][0] == 0x00080004 + 28


def test_single_position_kernels(oracle):
    rng = np.random.default_rng(1)
    src, refp = rnd(rng, 16, 64), rnd(rng, 16, 96)
    res = []
    for fn8, fn32 in ((ref.SadCalculation_8x8_16x16, ref.SadCalculation_32x32_64x64),
                      (oracle.svt_oracle_SadCalculation_8x8_16x16, oracle.svt_oracle_SadCalculation_32x32_64x64)):
        bs8, bm8 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        bs16, bm16 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        s16 = np.zeros(1, np.uint32)
        fn8(vp(P(src)), u32(64), vp(P(refp)), u32(96), vp(P(bs8)), vp(P(bs16)), vp(P(bm8)), vp(P(bm16)), u32(0x12340008), vp(P(s16)))
        all16 = (np.arange(16, dtype=np.uint32) * 37 + 11)
        bs32, bm32 = np.full(4, 64 * 64 * 255, np.uint32), np.zeros(4, np.uint32)
        bs64, bm64 = np.full(1, 64 * 64 * 255, np.uint32), np.zeros(1, np.uint32)
        fn32(vp(P(all16)), vp(P(bs32)), vp(P(bs64)), vp(P(bm32)), vp(P(bm64)), u32(0x12340008))
        res.append([x.tolist() for x in (bs8, bm8, bs16, bm16, s16, bs32, bm32, bs64, bm64)])
    assert res[0] == res[1]


@pytest.mark.parametrize("frac", [1, 2, 3])
def test_avc_interpolation(oracle, frac):
    rng = np.random.default_rng(frac)
    img = rnd(rng, 48, 96)
    for name in ("Horizontal", "Vertical"):
        outs = []
        for fn in (getattr(ref, "AvcStyleLumaInterpolationFilter" + name),
                   getattr(oracle, "svt_oracle_AvcStyleLumaInterpolationFilter" + name)):
            dst = np.zeros((32, 80), np.uint8)
            fn(vp(P(img) + 4 * 96 + 4), u32(96), vp(P(dst)), u32(80), u32(72), u32(30), vp(0), u32(frac))
            outs.append(dst)
        assert np.array_equal(outs[0], outs[1])


def test_average_sse_decimate(oracle):
    rng = np.random.default_rng(3)
    a, b = rnd(rng, 64, 64), rnd(rng, 64, 80)
    d0, d1 = np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8)
    ref.PictureAverageKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), vp(P(d0)), u32(64), u32(64), u32(64))
    oracle.svt_oracle_PictureAverageKernel(P(a), 64, P(b), 80, P(d1), 64, 64, 64)
    assert np.array_equal(d0, d1)
    ref.SpatialFullDistortionKernel.restype = u64
    for n in (4, 8, 16, 32, 64):
        assert ref.SpatialFullDistortionKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), u32(n), u32(n)) == \
            oracle.svt_oracle_SpatialFullDistortionKernel(P(a), 64, P(b), 80, n, n)
    big = rnd(rng, 64, 128)
    for step in (2, 4):
        o0, o1 = np.zeros((64 // step, 64), np.uint8), np.zeros((64 // step, 64), np.uint8)
        ref.Decimation2D(vp(P(big)), u32(128), u32(128), u32(64), vp(P(o0)), u32(64), u32(step))
        oracle.svt_oracle_Decimation2D(P(big), 128, 128, 64, P(o1), 64, step)
        assert np.array_equal(o0, o1)
