"""CPU-only: oracle restatement of the deblocking cores, SAO statistics/apply and 10-bit pack/unpack
against the reference's C_DEFAULT symbols (oracle/_ref/libsvtref.so)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
u32, i32, vp, u8 = C.c_uint32, C.c_int32, C.c_void_p, C.c_uint8


def P(a, off=0):
    return a.ctypes.data + off


def img(rng, bps, h, w, smooth=False):
    hi = 256 if bps == 1 else 1024
    a = rng.integers(0, hi, size=(h, w))
    if smooth:  # edges that actually pass the d < beta decision
        a = (hi // 2 + rng.integers(-3, 4, size=(h, w)) + 6 * (np.arange(w)[None, :] > w // 2)).clip(0, hi - 1)
    return a.astype(np.uint8 if bps == 1 else np.uint16)


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("vertical", [0, 1])
def test_luma_dlf_core(oracle, bps, vertical):
    oracle.svt_oracle_Luma4SampleEdgeDLFCore.argtypes = [C.c_int, vp, u32, C.c_int, i32, i32]
    fn = ref.Luma4SampleEdgeDLFCore if bps == 1 else ref.Luma4SampleEdgeDLFCore16bit
    rng = np.random.default_rng(bps * 2 + vertical)
    hits = 0
    for trial in range(200):
        a = img(rng, bps, 16, 16, smooth=trial % 4 != 0)
        if trial % 3 == 0 and not vertical:
            a = np.ascontiguousarray(a.T)
        tc, beta = int(rng.integers(0, 25)) << (2 if bps == 2 else 0), int(rng.integers(0, 65)) << (2 if bps == 2 else 0)
        w, g = a.copy(), a.copy()
        off = (8 * 16 + 8) * bps if vertical else (8 * 16 + 6) * bps
        fn(vp(P(w, off)), u32(16), u8(vertical), i32(tc), i32(beta))
        oracle.svt_oracle_Luma4SampleEdgeDLFCore(bps, P(g, off), 16, vertical, tc, beta)
        assert np.array_equal(w, g), trial
        hits += int(not np.array_equal(w, a))
    assert hits > 20  # the filter really modified samples in a good share of the trials


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("vertical", [0, 1])
def test_chroma_dlf_core(oracle, bps, vertical):
    oracle.svt_oracle_Chroma2SampleEdgeDLFCore.argtypes = [C.c_int, vp, vp, u32, C.c_int, u8, u8]
    fn = ref.Chroma2SampleEdgeDLFCore if bps == 1 else ref.Chroma2SampleEdgeDLFCore16bit
    rng = np.random.default_rng(bps + vertical)
    for trial in range(100):
        cb, cr = img(rng, bps, 8, 8, trial % 2 == 0), img(rng, bps, 8, 8, trial % 2 == 0)
        tcb, tcr = int(rng.integers(0, 25)), int(rng.integers(0, 25))
        wb, wr, gb, gr = cb.copy(), cr.copy(), cb.copy(), cr.copy()
        off = (4 * 8 + 4) * bps
        fn(vp(P(wb, off)), vp(P(wr, off)), u32(8), u8(vertical), u8(tcb), u8(tcr))
        oracle.svt_oracle_Chroma2SampleEdgeDLFCore(bps, P(gb, off), P(gr, off), 8, vertical, tcb, tcr)
        assert np.array_equal(wb, gb) and np.array_equal(wr, gr)


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("w,h", [(64, 64), (40, 56), (16, 8)])
def test_sao_gather(oracle, bps, w, h):
    oracle.svt_oracle_GatherSaoStatistics.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    rng = np.random.default_rng(bps + w)
    src = img(rng, bps, h, 80)
    rec = (src.astype(np.int32) + rng.integers(-9, 10, size=src.shape)).clip(0, 255 if bps == 1 else 1023).astype(src.dtype)
    full = ref.GatherSaoStatisticsLcuLossy_62x62 if bps == 1 else ref.GatherSaoStatisticsLcu_62x62_16bit
    eo = ref.GatherSaoStatisticsLcu_OnlyEo_90_45_135_Lossy if bps == 1 else ref.GatherSaoStatisticsLcu_62x62_OnlyEo_90_45_135_16bit
    for only in (0, 1):
        outs = []
        for which in (0, 1):
            bd, bc = np.full(32, 7, np.int32), np.full(32, 7, np.uint16)
            ed, ec = np.full((4, 5), 7, np.int32), np.full((4, 5), 7, np.uint16)
            if which == 0 and not only:
                full(vp(P(src)), u32(80), vp(P(rec)), u32(80), u32(w), u32(h), vp(P(bd)), vp(P(bc)), vp(P(ed)), vp(P(ec)))
            elif which == 0:
                eo(vp(P(src)), u32(80), vp(P(rec)), u32(80), u32(w), u32(h), vp(P(ed)), vp(P(ec)))
            else:
                oracle.svt_oracle_GatherSaoStatistics(bps, only, P(src), 80, P(rec), 80, w, h, P(bd), P(bc), P(ed), P(ec))
            outs.append((bd, bc, ed, ec))
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), (only, a, b)


@pytest.mark.parametrize("bps", [1, 2])
def test_sao_apply(oracle, bps):
    oracle.svt_oracle_SAOApplyBO.argtypes = [C.c_int, vp, u32, u32, vp, u32, u32]
    oracle.svt_oracle_SAOApplyEO.argtypes = [C.c_int, C.c_int, vp, u32, vp, vp, vp, u32, u32]
    sfx = "" if bps == 1 else "16bit"
    rng = np.random.default_rng(bps)
    for w, h in ((64, 64), (24, 40), (8, 8)):
        rec = img(rng, bps, h + 2, 80)
        off = np.array([-7, -3, 0, 4, 6], np.int8)
        for band in (0, 5, 28):
            a, b = rec.copy(), rec.copy()
            getattr(ref, "SAOApplyBO" + sfx)(vp(P(a)), u32(80), u32(band), vp(P(off)), u32(h), u32(w))
            oracle.svt_oracle_SAOApplyBO(bps, P(b), 80, band, P(off), h, w)
            assert np.array_equal(a, b)
        left = img(rng, bps, 1, h + 2)[0]
        upper = img(rng, bps, 1, w + 3)[0]
        names = {0: "SAOApplyEO_0", 1: "SAOApplyEO_90", 2: "SAOApplyEO_135", 3: "SAOApplyEO_45"}
        for t, name in names.items():
            a, b = rec.copy(), rec.copy()
            fn = getattr(ref, name + ("_16bit" if bps == 2 else ""))
            up = P(upper, bps)  # element 0 of `upper` is the top-left corner (index -1)
            if t == 0:
                fn(vp(P(a)), u32(80), vp(P(left)), vp(P(off)), u32(h), u32(w))
            elif t == 1:
                fn(vp(P(a)), u32(80), vp(up), vp(P(off)), u32(h), u32(w))
            else:
                fn(vp(P(a)), u32(80), vp(P(left)), vp(up), vp(P(off)), u32(h), u32(w))
            oracle.svt_oracle_SAOApplyEO(bps, t, P(b), 80, P(left), up, P(off), h, w)
            assert np.array_equal(a, b), name


def test_pack_unpack(oracle):
    rng = np.random.default_rng(0)
    w, h = 64, 24
    in8, inn = rng.integers(0, 256, (h, 80), np.uint8), rng.integers(0, 256, (h, 72), np.uint8)
    o0, o1 = np.zeros((h, 96), np.uint16), np.zeros((h, 96), np.uint16)
    ref.EB_ENC_msbPack2D(vp(P(in8)), u32(80), vp(P(inn)), vp(P(o0)), u32(72), u32(96), u32(w), u32(h))
    oracle.svt_oracle_msbPack2D(vp(P(in8)), u32(80), vp(P(inn)), vp(P(o1)), u32(72), u32(96), u32(w), u32(h))
    assert np.array_equal(o0, o1)
    o0[:], o1[:] = 0, 0
    ref.CompressedPackmsb(vp(P(in8)), u32(80), vp(P(inn)), vp(P(o0)), u32(72), u32(96), u32(w), u32(h))
    oracle.svt_oracle_CompressedPackmsb(vp(P(in8)), u32(80), vp(P(inn)), vp(P(o1)), u32(72), u32(96), u32(w), u32(h))
    assert np.array_equal(o0, o1)
    c0, c1 = np.zeros((h, 32), np.uint8), np.zeros((h, 32), np.uint8)
    ref.CPack_C(vp(P(inn)), u32(72), vp(P(c0)), u32(32), None, u32(w), u32(h))
    oracle.svt_oracle_CPack(vp(P(inn)), u32(72), vp(P(c1)), u32(32), u32(w), u32(h))
    assert np.array_equal(c0, c1)
    in16 = rng.integers(0, 1024, (h, 80)).astype(np.uint16)
    a8, an, b8, bn = (np.zeros((h, 72), np.uint8) for _ in range(4))
    ref.EB_ENC_msbUnPack2D(vp(P(in16)), u32(80), vp(P(a8)), vp(P(an)), u32(72), u32(72), u32(w), u32(h))
    oracle.svt_oracle_msbUnPack2D(vp(P(in16)), u32(80), vp(P(b8)), vp(P(bn)), u32(72), u32(72), u32(w), u32(h))
    assert np.array_equal(a8, b8) and np.array_equal(an, bn)
    a8[:], b8[:] = 0, 0
    ref.UnPack8BitData(vp(P(in16)), u32(80), vp(P(a8)), u32(72), u32(w), u32(h))
    oracle.svt_oracle_msbUnPack2D(vp(P(in16)), u32(80), vp(P(b8)), None, u32(72), u32(0), u32(w), u32(h))
    assert np.array_equal(a8, b8)
    in16b = rng.integers(0, 1024, (h, 96)).astype(np.uint16)
    ref.UnpackAvg(vp(P(in16)), u32(80), vp(P(in16b)), u32(96), vp(P(a8)), u32(72), u32(w), u32(h))
    oracle.svt_oracle_UnpackAvg(vp(P(in16)), u32(80), vp(P(in16b)), u32(96), vp(P(b8)), u32(72), u32(w), u32(h))
    assert np.array_equal(a8, b8)
