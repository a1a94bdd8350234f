"""-m gpu: EncDec leaf families (transforms, quantisation, distortion, SATD, residual/addition)
through the C-ABI - per-call leaf entry points and the batched device-pointer forms - against
the oracle (itself pinned to the reference's C_DEFAULT symbols in tests/test_oracle_txfm.py)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

pytestmark = pytest.mark.gpu
u32, u64, i32, vp = C.c_uint32, C.c_uint64, C.c_int32, C.c_void_p

FWD = {(0, 32): "Transform32x32", (1, 32): "Transform32x32Estimate", (0, 16): "Transform16x16",
       (1, 16): "Transform16x16Estimate", (0, 8): "Transform8x8", (0, 4): "Transform4x4", (2, 4): "DstTransform4x4"}
INV = {(0, 32): "InvTransform32x32", (0, 16): "InvTransform16x16", (0, 8): "InvTransform8x8",
       (0, 4): "InvTransform4x4", (2, 4): "InvDstTransform4x4"}


def P(a):
    return a.ctypes.data


@pytest.fixture(scope="module")
def libs(product, oracle):
    oracle.svt_oracle_FwdTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    oracle.svt_oracle_InvTransform.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    oracle.svt_oracle_QuantizeInvQuantize.argtypes = [vp, u32, vp, vp, u32, u32, i32, i32, i32, i32, u32, vp]
    oracle.svt_oracle_FullDistortionKernel_32bit.argtypes = [vp, u32, vp, u32, vp, u32, u32, C.c_int]
    for n in ("Compute8x8Satd", "Compute4x4Satd", "Compute8x8Satd_U8", "Compute4x4Satd_U8"):
        getattr(oracle, "svt_oracle_" + n).restype = u64
        getattr(product, "svt_amd_" + n).restype = u64
    for n in ("fwd_transform_batch", "inv_transform_batch"):
        getattr(product, "svt_amd_" + n).argtypes = [vp, C.c_int, C.c_int, u32, vp, vp, u32]
    product.svt_amd_quantize_batch.argtypes = [vp, C.c_int, u32, u32, i32, i32, i32, i32, vp, vp, vp, vp, u32]
    product.svt_amd_full_distortion_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, u32]
    product.svt_amd_satd_batch.argtypes = [vp, C.c_int, vp, vp, u32]
    return product, oracle


def sample(size, seed, scale):
    rng = np.random.default_rng(seed)
    return rng.integers(-scale, scale + 1, size=(size, 64)).astype(np.int16)


@pytest.mark.parametrize("kind,size", sorted(FWD))
def test_forward_leaf(libs, kind, size):
    product, oracle = libs
    for inc, scale in ((0, 255), (2, 1023), (0, 32767)):
        src = sample(size, size + kind + scale, scale)
        want, got = np.zeros((size, 48), np.int16), np.zeros((size, 48), np.int16)
        oracle.svt_oracle_FwdTransform(kind, size, P(src), 64, P(want), 48, None, inc)
        getattr(product, "svt_amd_" + FWD[(kind, size)])(vp(P(src)), u32(64), vp(P(got)), u32(48), None, u32(inc))
        assert np.array_equal(got, want), (FWD[(kind, size)], inc, scale)


@pytest.mark.parametrize("kind,size", sorted(INV))
def test_inverse_leaf(libs, kind, size):
    product, oracle = libs
    for inc, scale in ((0, 400), (2, 4000), (0, 32767)):
        src = sample(size, size + kind + scale + 7, scale)
        want, got = np.zeros((size, 48), np.int16), np.zeros((size, 48), np.int16)
        oracle.svt_oracle_InvTransform(kind, size, P(src), 64, P(want), 48, None, inc)
        getattr(product, "svt_amd_" + INV[(kind, size)])(vp(P(src)), u32(64), vp(P(got)), u32(48), None, u32(inc))
        assert np.array_equal(got, want), (INV[(kind, size)], inc, scale)


def qparams(size, qp):
    qf = [26214, 23302, 20560, 18396, 16384, 14564][qp % 6]
    ff = [40, 45, 51, 57, 64, 72][qp % 6]
    lg = int(np.log2(size))
    qbits = 14 + qp // 6 + (15 - 8 - lg)
    shift_num = 20 - 14 - (15 - 8 - lg)
    return qf, 171 << (qbits - 9), qbits, ff << (qp // 6), 1 << (shift_num - 1), shift_num


@pytest.mark.parametrize("size", [4, 8, 16, 32])
def test_quant_distortion_satd_leaf(libs, size):
    product, oracle = libs
    for qp, scale in ((22, 300), (37, 32767)):
        coeff = sample(size, qp + size, scale)
        qf, qo, qb, ffs, iqo, sn = qparams(size, qp)
        outs = []
        for fn in (oracle.svt_oracle_QuantizeInvQuantize, product.svt_amd_QuantizeInvQuantize):
            q, r, nz = np.zeros((size, 64), np.int16), np.zeros((size, 64), np.int16), u32(9)
            fn(vp(P(coeff)), u32(64), vp(P(q)), vp(P(r)), u32(qf), u32(qo), i32(qb), i32(ffs), i32(iqo), i32(sn), u32(size),
               C.byref(nz))
            outs.append((q.copy(), r.copy(), nz.value))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2]
        rec = outs[0][1]
        for mode, name in ((0, "FullDistortionKernel_32bit"), (1, "FullDistortionKernelCbfZero_32bit"),
                           (2, "FullDistortionKernelIntra_32bit")):
            w, g = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
            oracle.svt_oracle_FullDistortionKernel_32bit(P(coeff), 64, P(rec), 64, P(w), size, size, mode)
            getattr(product, "svt_amd_" + name)(vp(P(coeff)), u32(64), vp(P(rec)), u32(64), vp(P(g)), u32(size), u32(size))
            assert np.array_equal(w, g), name
    rng = np.random.default_rng(size)
    for scale in (255, 32767):
        d8 = rng.integers(-scale, scale + 1, 64).astype(np.int16)
        d4 = rng.integers(-scale, scale + 1, 16).astype(np.int16)
        assert product.svt_amd_Compute8x8Satd(vp(P(d8))) == oracle.svt_oracle_Compute8x8Satd(vp(P(d8)))
        assert product.svt_amd_Compute4x4Satd(vp(P(d4))) == oracle.svt_oracle_Compute4x4Satd(vp(P(d4)))
    img = rng.integers(0, 256, (16, 40), np.uint8)
    for n in ("Compute8x8Satd_U8", "Compute4x4Satd_U8"):
        a, b = u64(5), u64(5)
        assert getattr(product, "svt_amd_" + n)(vp(P(img) + 3), C.byref(a), u32(40)) == \
            getattr(oracle, "svt_oracle_" + n)(vp(P(img) + 3), C.byref(b), u32(40))
        assert a.value == b.value


def test_residual_addition_leaf(libs):
    product, oracle = libs
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, 256, (64, 64), np.uint8), rng.integers(0, 256, (64, 80), np.uint8)
    res = rng.integers(-400, 400, (64, 64)).astype(np.int16)
    for w, h in ((4, 4), (8, 8), (32, 32), (64, 64), (24, 12)):
        r0, r1 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
        oracle.svt_oracle_ResidualKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), vp(P(r0)), u32(64), u32(w), u32(h))
        product.svt_amd_ResidualKernel(vp(P(a)), u32(64), vp(P(b)), u32(80), vp(P(r1)), u32(64), u32(w), u32(h))
        assert np.array_equal(r0, r1)
        o0, o1 = np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8)
        oracle.svt_oracle_PictureAdditionKernel(vp(P(b)), u32(80), vp(P(res)), u32(64), vp(P(o0)), u32(64), u32(w), u32(h))
        product.svt_amd_PictureAdditionKernel(vp(P(b)), u32(80), vp(P(res)), u32(64), vp(P(o1)), u32(64), u32(w), u32(h))
        assert np.array_equal(o0, o1)


@pytest.mark.parametrize("kind,size", sorted(FWD))
def test_batched_pipeline_matches_oracle(libs, gpu_ctx, kind, size):
    """residual blocks -> forward -> quant/iquant -> distortion -> inverse, 257 blocks per launch,
    device pointers only; every block compared with the oracle."""
    import torch
    product, oracle = libs
    n = 257
    rng = np.random.default_rng(size * 3 + kind)
    res = rng.integers(-255, 256, size=(n, size, size)).astype(np.int16)
    res[0] = 255
    res[1] = rng.integers(-32768, 32768, size=(size, size))
    dev = torch.device("cuda", 0)
    d_res = torch.from_numpy(res).to(dev)
    d_coef, d_q, d_rec, d_inv = (torch.empty_like(d_res) for _ in range(4))
    d_nz = torch.empty(n, dtype=torch.int32, device=dev)
    d_dist = torch.empty((n, 2), dtype=torch.int64, device=dev)
    d_satd = torch.empty(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    qf, qo, qb, ffs, iqo, sn = qparams(size, 30)
    ikind = 2 if kind == 2 else 0
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    assert product.svt_amd_fwd_transform_batch(gpu_ctx, kind, size, 0, d_res.data_ptr(), d_coef.data_ptr(), n) == 0
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    assert product.svt_amd_quantize_batch(gpu_ctx, size, qf, qo, qb, ffs, iqo, sn, d_coef.data_ptr(), d_q.data_ptr(),
                                          d_rec.data_ptr(), d_nz.data_ptr(), n) == 0
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    assert product.svt_amd_full_distortion_batch(gpu_ctx, size, 0, d_coef.data_ptr(), d_rec.data_ptr(), d_dist.data_ptr(), n) == 0
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    assert product.svt_amd_inv_transform_batch(gpu_ctx, ikind, size, 0, d_rec.data_ptr(), d_inv.data_ptr(), n) == 0
    if size in (4, 8):
        torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
        assert product.svt_amd_satd_batch(gpu_ctx, size, d_res.data_ptr(), d_satd.data_ptr(), n) == 0
    assert product.svt_amd_synchronize(gpu_ctx) == 0
    coef, q, rec, inv = (x.cpu().numpy() for x in (d_coef, d_q, d_rec, d_inv))
    nz, dist, satd = d_nz.cpu().numpy(), d_dist.cpu().numpy(), d_satd.cpu().numpy()
    for b in range(n):
        wc, wq, wr, wi = (np.zeros((size, size), np.int16) for _ in range(4))
        wnz, wd = u32(0), np.zeros(2, np.uint64)
        src = np.ascontiguousarray(res[b])
        oracle.svt_oracle_FwdTransform(kind, size, P(src), size, P(wc), size, None, 0)
        oracle.svt_oracle_QuantizeInvQuantize(P(wc), size, P(wq), P(wr), qf, qo, qb, ffs, iqo, sn, size, C.byref(wnz))
        oracle.svt_oracle_FullDistortionKernel_32bit(P(wc), size, P(wr), size, P(wd), size, size, 0)
        oracle.svt_oracle_InvTransform(ikind, size, P(wr), size, P(wi), size, None, 0)
        assert np.array_equal(coef[b], wc) and np.array_equal(q[b], wq) and np.array_equal(rec[b], wr), b
        assert np.array_equal(inv[b], wi), b
        assert int(nz[b]) == wnz.value and dist[b].astype(np.uint64).tolist() == wd.tolist(), b
        if size == 8:
            assert int(satd[b]) == oracle.svt_oracle_Compute8x8Satd(vp(P(src))), b
        if size == 4:
            assert int(satd[b]) == oracle.svt_oracle_Compute4x4Satd(vp(P(src))), b


# ---- 32x32 transforms on the matrix cores (v_mfma_i32_32x32x32_i8, txfm_mfma.hip) ------------------------------------------
@pytest.mark.parametrize("kind,inc,scale", [(0, 0, 255), (0, 2, 1023), (0, 0, 32767), (1, 0, 255), (1, 0, 8191), (1, 0, 32767), (1, 2, 1023)])
def test_forward_32x32_mfma_matches_oracle(libs, gpu_ctx, kind, inc, scale):
    """Full-precision and Estimate forward DCT as integer-MFMA products, all of int16's range: 8-bit / 10-bit residuals (the encoder's
    inputs), the edge of the Estimate form's wrap-free domain, and +-32767 (where its 16-bit butterfly levels wrap in the reference:
    the kernel flags those blocks and the VALU butterfly redoes them)."""
    import torch
    product, oracle = libs
    product.svt_amd_fwd_transform_mfma_batch.argtypes = [vp, C.c_int, C.c_int, u32, vp, vp, u32]
    rng = np.random.default_rng(kind * 100 + scale)
    n = 301
    blocks = rng.integers(-scale, scale + 1, size=(n, 32, 32)).astype(np.int16)
    blocks[0] = scale            # flat extremes
    blocks[1] = -scale
    blocks[2, :, ::2] = scale    # alternating columns: large odd-part sums
    blocks[2, :, 1::2] = -scale
    blocks[3] = 0
    blocks[4, 5, 7] = scale      # a single sample
    if scale == 32767:           # mixed batch: most blocks stay inside the wrap-free domain
        blocks[10:200] = rng.integers(-255, 256, size=(190, 32, 32))
    d_in = torch.from_numpy(blocks).cuda()
    d_out = torch.full((n, 32, 32), 12345, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    rc = product.svt_amd_fwd_transform_mfma_batch(gpu_ctx, kind, 32, inc, d_in.data_ptr(), d_out.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    got = d_out.cpu().numpy()
    for b in range(n):
        want = np.zeros((32, 32), np.int16)
        oracle.svt_oracle_FwdTransform(kind, 32, P(np.ascontiguousarray(blocks[b])), 32, P(want), 32, None, inc)
        assert np.array_equal(got[b], want), (kind, inc, scale, b)


@pytest.mark.parametrize("inc,scale", [(0, 400), (2, 4000), (0, 32767)])
def test_inverse_32x32_mfma_matches_oracle(libs, gpu_ctx, inc, scale):
    import torch
    product, oracle = libs
    product.svt_amd_inv_transform_mfma_batch.argtypes = [vp, C.c_int, u32, vp, vp, u32]
    rng = np.random.default_rng(scale + inc)
    n = 203
    blocks = rng.integers(-scale, scale + 1, size=(n, 32, 32)).astype(np.int16)
    blocks[0] = scale
    blocks[1] = -scale
    blocks[2] = 0
    blocks[2, 0, 0] = scale      # DC only
    blocks[3, 1::2, :] = 0       # even rows only
    d_in = torch.from_numpy(blocks).cuda()
    d_out = torch.full((n, 32, 32), 12345, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    rc = product.svt_amd_inv_transform_mfma_batch(gpu_ctx, 32, inc, d_in.data_ptr(), d_out.data_ptr(), n)
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    got = d_out.cpu().numpy()
    for b in range(n):
        want = np.zeros((32, 32), np.int16)
        oracle.svt_oracle_InvTransform(0, 32, P(np.ascontiguousarray(blocks[b])), 32, P(want), 32, None, inc)
        assert np.array_equal(got[b], want), (inc, scale, b)
