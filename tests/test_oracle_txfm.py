"""CPU-only: pin the oracle's transform / quantisation / distortion / SATD restatements against the
REFERENCE's C_DEFAULT symbols in oracle/_ref/libsvtref.so."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
u32, u64, i32, vp = C.c_uint32, C.c_uint64, C.c_int32, C.c_void_p


def P(a):
    return a.ctypes.data


FWD = {(0, 32): "Transform32x32", (1, 32): "Transform32x32Estimate", (0, 16): "Transform16x16",
       (1, 16): "Transform16x16Estimate", (0, 8): "Transform8x8", (0, 4): "Transform4x4", (2, 4): "DstTransform4x4"}
INV = {(0, 32): "InvTransform32x32", (0, 16): "InvTransform16x16", (0, 8): "InvTransform8x8",
       (0, 4): "InvTransform4x4", (2, 4): "InvDstTransform4x4"}


def blocks(size, seed):
    rng = np.random.default_rng(seed)
    yield rng.integers(-255, 256, size=(size, 64)).astype(np.int16)          # 8-bit residuals
    yield rng.integers(-1023, 1024, size=(size, 64)).astype(np.int16)        # 10-bit residuals
    yield rng.integers(-32768, 32768, size=(size, 64)).astype(np.int16)      # full range: exercises the 16-bit wraps
    yield np.full((size, 64), 255, np.int16)
    yield np.full((size, 64), -32768, np.int16)


@pytest.mark.parametrize("kind,size", sorted(FWD))
@pytest.mark.parametrize("inc", [0, 2])
def test_forward_transforms(oracle, kind, size, inc):
    oracle.svt_oracle_FwdTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    for src in blocks(size, size * 10 + kind):
        want = np.zeros((size, 48), np.int16)
        got = np.zeros((size, 48), np.int16)
        inner = np.zeros(32 * 32, np.int16)
        getattr(ref, FWD[(kind, size)])(vp(P(src)), u32(64), vp(P(want)), u32(48), vp(P(inner)), u32(inc))
        oracle.svt_oracle_FwdTransform(kind, size, P(src), 64, P(got), 48, None, inc)
        assert np.array_equal(got, want), (FWD[(kind, size)], inc)


@pytest.mark.parametrize("kind,size", sorted(INV))
@pytest.mark.parametrize("inc", [0, 2])
def test_inverse_transforms(oracle, kind, size, inc):
    oracle.svt_oracle_InvTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    for src in blocks(size, size * 10 + kind + 100):
        want = np.zeros((size, 48), np.int16)
        got = np.zeros((size, 48), np.int16)
        inner = np.zeros(32 * 32, np.int16)
        getattr(ref, INV[(kind, size)])(vp(P(src)), u32(64), vp(P(want)), u32(48), vp(P(inner)), u32(inc))
        oracle.svt_oracle_InvTransform(kind, size, P(src), 64, P(got), 48, None, inc)
        assert np.array_equal(got, want), (INV[(kind, size)], inc)


@pytest.mark.parametrize("size", [4, 8, 16, 32])
@pytest.mark.parametrize("qp", [10, 22, 32, 45])
def test_quantize_inv_quantize(oracle, size, qp):
    oracle.svt_oracle_QuantizeInvQuantize.argtypes = [vp, u32, vp, vp, u32, u32, i32, i32, i32, i32, u32, vp]
    rng = np.random.default_rng(size + qp)
    qf = [26214, 23302, 20560, 18396, 16384, 14564][qp % 6]
    ff = [40, 45, 51, 57, 64, 72][qp % 6]
    lg = int(np.log2(size))
    qbits = 14 + qp // 6 + (15 - 8 - lg)          # QUANT_SHIFT + qp/6 + transform shift (8-bit)
    qoff = 171 << (qbits - 9)
    shift_num = 20 - 14 - (15 - 8 - lg)            # IQUANT_SHIFT - QUANT_IQUANT_SHIFT - transform shift
    ffs, iqo = ff << (qp // 6), 1 << (shift_num - 1)
    for scale in (40, 4000, 32767):
        coeff = rng.integers(-scale, scale + 1, size=(size, 64)).astype(np.int16)
        outs = []
        for fn in (ref.QuantizeInvQuantize, oracle.svt_oracle_QuantizeInvQuantize):
            q, r, nz = np.zeros((size, 64), np.int16), np.zeros((size, 64), np.int16), u32(7)
            fn(vp(P(coeff)), u32(64), vp(P(q)), vp(P(r)), u32(qf), u32(qoff), i32(qbits), i32(ffs), i32(iqo),
               i32(shift_num), u32(size), C.byref(nz))
            outs.append((q.copy(), r.copy(), nz.value))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2]


def test_update_qiq_coef(oracle):
    oracle.svt_oracle_UpdateQiQCoef.argtypes = [vp, vp, u32, i32, i32, i32, u32, vp, u32, u32, u32, u32, C.c_uint8]
    for nz0 in (0, 3, 12):
        for slice_type in (0, 1, 2):
            for cb in (0, 1):
                outs = []
                for fn in (ref.UpdateQiQCoef, oracle.svt_oracle_UpdateQiQCoef):
                    q, r = np.zeros((16, 32), np.int16), np.zeros((16, 32), np.int16)
                    if nz0:
                        q[0, :nz0] = 5
                    nz = u32(nz0)
                    fn(vp(P(q)), vp(P(r)), u32(32), i32(64 << 3), i32(1 << 1), i32(2), u32(16), C.byref(nz), u32(0),
                       u32(slice_type), u32(0), u32(cb), C.c_uint8(1))
                    outs.append((q.copy(), r.copy(), nz.value))
                assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
                assert outs[0][2] == outs[1][2]


def test_residual_addition_zero(oracle):
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, 256, (64, 64), np.uint8), rng.integers(0, 256, (64, 80), np.uint8)
    for w, h in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64)):
        r0, r1 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
        ref.ResidualKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), vp(P(r0)), u32(64), u32(w), u32(h))
        oracle.svt_oracle_ResidualKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), vp(P(r1)), u32(64), u32(w), u32(h))
        assert np.array_equal(r0, r1)
        res = rng.integers(-400, 400, (64, 64)).astype(np.int16)
        o0, o1 = np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8)
        ref.PictureAdditionKernel(vp(P(b)), u32(80), vp(P(res)), u32(64), vp(P(o0)), u32(64), u32(w), u32(h))
        oracle.svt_oracle_PictureAdditionKernel(vp(P(b)), u32(80), vp(P(res)), u32(64), vp(P(o1)), u32(64), u32(w), u32(h))
        assert np.array_equal(o0, o1)
        z0, z1 = res.copy(), res.copy()
        ref.ZeroOutCoeffKernel(vp(P(z0)), u32(64), u32(130), u32(w // 2), u32(h // 2))
        oracle.svt_oracle_ZeroOutCoeffKernel(vp(P(z1)), u32(64), u32(130), u32(w // 2), u32(h // 2))
        assert np.array_equal(z0, z1)


@pytest.mark.parametrize("mode,name", [(0, "FullDistortionKernel_32bit"), (1, "FullDistortionKernelCbfZero_32bit"),
                                       (2, "FullDistortionKernelIntra_32bit")])
def test_full_distortion(oracle, mode, name):
    oracle.svt_oracle_FullDistortionKernel_32bit.argtypes = [vp, u32, vp, u32, vp, u32, u32, C.c_int]
    rng = np.random.default_rng(mode)
    for scale in (300, 32767):
        c = rng.integers(-scale, scale + 1, (32, 64)).astype(np.int16)
        r = rng.integers(-scale, scale + 1, (32, 48)).astype(np.int16)
        for n in (4, 8, 16, 32):
            w, g = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
            getattr(ref, name)(vp(P(c)), u32(64), vp(P(r)), u32(48), vp(P(w)), u32(n), u32(n))
            oracle.svt_oracle_FullDistortionKernel_32bit(P(c), 64, P(r), 48, P(g), n, n, mode)
            assert np.array_equal(w, g), (name, n, scale)


def test_satd(oracle):
    for f in (oracle.svt_oracle_Compute8x8Satd, oracle.svt_oracle_Compute4x4Satd, oracle.svt_oracle_Compute8x8Satd_U8,
              oracle.svt_oracle_Compute4x4Satd_U8, ref.Compute8x8Satd, ref.Compute4x4Satd, ref.Compute8x8Satd_U8,
              ref.Compute4x4Satd_U8):
        f.restype = u64
    rng = np.random.default_rng(4)
    for scale in (255, 1023, 32767):
        d8 = rng.integers(-scale, scale + 1, 64).astype(np.int16)
        d4 = rng.integers(-scale, scale + 1, 16).astype(np.int16)
        assert ref.Compute8x8Satd(vp(P(d8))) == oracle.svt_oracle_Compute8x8Satd(vp(P(d8)))
        assert ref.Compute4x4Satd(vp(P(d4))) == oracle.svt_oracle_Compute4x4Satd(vp(P(d4)))
    img = rng.integers(0, 256, (16, 40), np.uint8)
    for n, fr, fo in ((8, ref.Compute8x8Satd_U8, oracle.svt_oracle_Compute8x8Satd_U8),
                      (4, ref.Compute4x4Satd_U8, oracle.svt_oracle_Compute4x4Satd_U8)):
        a, b = u64(11), u64(11)
        assert fr(vp(P(img) + 3), C.byref(a), u32(40)) == fo(vp(P(img) + 3), C.byref(b), u32(40))
        assert a.value == b.value
