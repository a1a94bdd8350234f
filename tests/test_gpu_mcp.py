"""-m gpu: HEVC motion-compensation interpolation through the C-ABI (78 leaf entry points + batched forms) vs the
oracle (pinned to the reference's tables in tests/test_oracle_mcp.py)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_mcp import SIZES, decl, refplane

pytestmark = pytest.mark.gpu
u32, vp, i32 = C.c_uint32, C.c_void_p, C.c_int32
POS = "abcdefghijknpqr"
FRAC = {p: ((i + 1) & 3, (i + 1) >> 2) for i, p in enumerate(POS)}

MCP_BLOCK = np.dtype([("ref_off", "<i4"), ("dst_off", "<i4"), ("w", "<u2"), ("h", "<u2"), ("fx", "u1"), ("fy", "u1"),
                      ("pad", "u1", 2)])
BI_BLOCK = np.dtype([("l0_off", "<i4"), ("l1_off", "<i4"), ("dst_off", "<i4"), ("w", "<u2"), ("h", "<u2")])


def test_struct_sizes():
    assert MCP_BLOCK.itemsize == 16 and BI_BLOCK.itemsize == 16


@pytest.mark.parametrize("bps", [1, 2])
def test_luma_leaf(product, oracle, bps):
    decl(oracle)
    rng = np.random.default_rng(bps)
    dt = np.uint8 if bps == 1 else np.uint16
    sfx = "" if bps == 1 else "16bit"
    names = [("LumaInterpolationCopy" + sfx, "LumaInterpolationCopyOutRaw" + sfx, 0, 0)] + \
            [("LumaInterpolationFilterPos%sNew%s" % (p, sfx), "LumaInterpolationFilterPos%sOutRaw%s" % (p, sfx)) + FRAC[p] for p in POS]
    for uni, raw, fx, fy in names:
        for k, (w, h) in enumerate(SIZES[:4] if fx and fy else SIZES):
            plane = refplane(rng, bps, k == 1)
            base = plane.ctypes.data + (16 * 112 + 16) * bps
            want, got = np.full((h, 80), 7, dt), np.full((h, 80), 7, dt)
            getattr(product, "svt_amd_" + uni)(vp(base), u32(112), vp(got.ctypes.data), u32(80), u32(w), u32(h), None)
            oracle.svt_oracle_mcp(bps, 0, 0, fx, fy, base, 112, want.ctypes.data, 80, w, h)
            assert np.array_equal(want, got), (uni, w, h)
            want, got = np.full(h * w, 7, np.int16), np.full(h * w, 7, np.int16)
            getattr(product, "svt_amd_" + raw)(vp(base), u32(112), vp(got.ctypes.data), u32(w), u32(h), None)
            oracle.svt_oracle_mcp(bps, 0, 1, fx, fy, base, 112, want.ctypes.data, 0, w, h)
            assert np.array_equal(want, got), (raw, w, h)


@pytest.mark.parametrize("bps", [1, 2])
def test_chroma_leaf(product, oracle, bps):
    decl(oracle)
    rng = np.random.default_rng(10 + bps)
    dt = np.uint8 if bps == 1 else np.uint16
    sfx = "" if bps == 1 else "16bit"
    for fx, fy in [(0, 0), (3, 0), (0, 5), (7, 0), (1, 1), (4, 4), (7, 2), (2, 7), (6, 5)]:
        kind = "Copy" if not (fx or fy) else ("FilterOneD" if not (fx and fy) else "FilterTwoD")
        for k, (w, h) in enumerate([(4, 4), (8, 4), (16, 16), (32, 32), (4, 16)]):
            plane = refplane(rng, bps, k == 1)
            base = plane.ctypes.data + (16 * 112 + 16) * bps
            want, got = np.full((h, 48), 7, dt), np.full((h, 48), 7, dt)
            getattr(product, "svt_amd_ChromaInterpolation%s%s" % (kind, sfx))(
                vp(base), u32(112), vp(got.ctypes.data), u32(48), u32(w), u32(h), None, u32(fx), u32(fy))
            oracle.svt_oracle_mcp(bps, 1, 0, fx, fy, base, 112, want.ctypes.data, 48, w, h)
            assert np.array_equal(want, got), (kind, fx, fy, w, h)
            want, got = np.full(h * w, 7, np.int16), np.full(h * w, 7, np.int16)
            getattr(product, "svt_amd_ChromaInterpolation%sOutRaw%s" % (kind, sfx))(
                vp(base), u32(112), vp(got.ctypes.data), u32(w), u32(h), None, u32(fx), u32(fy))
            oracle.svt_oracle_mcp(bps, 1, 1, fx, fy, base, 112, want.ctypes.data, 0, w, h)
            assert np.array_equal(want, got), (kind, "raw", fx, fy, w, h)


def test_bipred_leaf(product, oracle):
    decl(oracle)
    rng = np.random.default_rng(3)
    for w, h in SIZES:
        l0 = rng.integers(-8192, 8192, h * w).astype(np.int16)
        l1 = rng.integers(-8192, 8192, h * w).astype(np.int16)
        for offset in (64 + 16384, 64):
            want, got = np.full((h, 80), 7, np.uint8), np.full((h, 80), 7, np.uint8)
            product.svt_amd_BiPredClipping(u32(w), u32(h), vp(l0.ctypes.data), vp(l1.ctypes.data), vp(got.ctypes.data), u32(80), i32(offset))
            oracle.svt_oracle_BiPredClipping(1, w, h, l0.ctypes.data, l1.ctypes.data, want.ctypes.data, 80, offset)
            assert np.array_equal(want, got)
        want, got = np.full((h, 80), 7, np.uint16), np.full((h, 80), 7, np.uint16)
        product.svt_amd_BiPredClipping16bit(u32(w), u32(h), vp(l0.ctypes.data), vp(l1.ctypes.data), vp(got.ctypes.data), u32(80))
        oracle.svt_oracle_BiPredClipping(2, w, h, l0.ctypes.data, l1.ctypes.data, want.ctypes.data, 80, 0)
        assert np.array_equal(want, got)


@pytest.mark.parametrize("bps", [1, 2])
def test_mcp_batch_picture(product, gpu_ctx, oracle, bps):
    """A 1080p-sized plane predicted as 16x16 PUs with per-PU motion vectors in one launch (uni), then the same PUs as
    bi-prediction (two raw launches + one clipping launch); checked against the oracle PU by PU on a sample of PUs and
    through a property on all of them: zero-fraction prediction == the reference plane itself."""
    import torch
    decl(oracle)
    product.svt_amd_mcp_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, u32, vp, u32, vp, u32]
    product.svt_amd_bipred_clip_batch.argtypes = [vp, C.c_int, vp, vp, vp, u32, i32, vp, u32]
    rng = np.random.default_rng(bps)
    W, H, PADX = 1920, 1080, 80
    hi = 256 if bps == 1 else 1024
    dt = np.uint8 if bps == 1 else np.uint16
    ref = rng.integers(0, hi, (H + 2 * PADX, W + 2 * PADX)).astype(dt)
    st = W + 2 * PADX
    d_ref = torch.from_numpy(ref.view(np.uint8).copy()).cuda()
    nx, ny = W // 16, H // 16
    blocks = np.zeros(nx * ny, MCP_BLOCK)
    mvx, mvy = rng.integers(-64 * 4, 64 * 4, nx * ny), rng.integers(-64 * 4, 64 * 4, nx * ny)
    mvx[::7], mvy[::7] = (mvx[::7] >> 2) << 2, (mvy[::7] >> 2) << 2  # some integer vectors
    bx, by = np.tile(np.arange(nx) * 16, ny), np.repeat(np.arange(ny) * 16, nx)
    blocks["ref_off"] = (PADX + by + (mvy >> 2)) * st + PADX + bx + (mvx >> 2)
    blocks["dst_off"] = by * W + bx
    blocks["w"], blocks["h"], blocks["fx"], blocks["fy"] = 16, 16, mvx & 3, mvy & 3
    d_blocks = torch.from_numpy(blocks.view(np.uint8).copy()).cuda()
    d_dst = torch.zeros(W * H * bps, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    rc = product.svt_amd_mcp_batch(gpu_ctx, bps, 0, 0, d_ref.data_ptr(), st, d_dst.data_ptr(), W, d_blocks.data_ptr(), len(blocks))
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    pred = d_dst.cpu().numpy().view(dt).reshape(H, W)
    flat = ref.reshape(-1)
    for i in list(range(0, len(blocks), 97)) + [len(blocks) - 1]:
        b = blocks[i]
        want = np.zeros((16, 16), dt)
        oracle.svt_oracle_mcp(bps, 0, 0, int(b["fx"]), int(b["fy"]), flat.ctypes.data + int(b["ref_off"]) * bps, st,
                              want.ctypes.data, 16, 16, 16)
        y, x = divmod(int(b["dst_off"]), W)
        assert np.array_equal(pred[y:y + 16, x:x + 16], want), i
    # property over ALL PUs with integer vectors: prediction is a plain copy of the displaced reference block
    for i in np.nonzero((blocks["fx"] == 0) & (blocks["fy"] == 0))[0]:
        y, x = divmod(int(blocks[i]["dst_off"]), W)
        ry, rx = divmod(int(blocks[i]["ref_off"]), st)
        assert np.array_equal(pred[y:y + 16, x:x + 16], ref[ry:ry + 16, rx:rx + 16])

    # bi-prediction: raw list 0 + raw list 1 (second vector field) -> clipping
    blocks1 = blocks.copy()
    mvx1, mvy1 = rng.integers(-32 * 4, 32 * 4, nx * ny), rng.integers(-32 * 4, 32 * 4, nx * ny)
    blocks1["ref_off"] = (PADX + by + (mvy1 >> 2)) * st + PADX + bx + (mvx1 >> 2)
    blocks1["fx"], blocks1["fy"] = mvx1 & 3, mvy1 & 3
    raw_blocks0, raw_blocks1 = blocks.copy(), blocks1
    raw_blocks0["dst_off"] = np.arange(nx * ny) * 256
    raw_blocks1["dst_off"] = np.arange(nx * ny) * 256
    d_raw0 = torch.zeros(nx * ny * 256, dtype=torch.int16, device="cuda")
    d_raw1 = torch.zeros(nx * ny * 256, dtype=torch.int16, device="cuda")
    for rb, d_raw in ((raw_blocks0, d_raw0), (raw_blocks1, d_raw1)):
        d_b = torch.from_numpy(rb.view(np.uint8).copy()).cuda()
        torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
        assert product.svt_amd_mcp_batch(gpu_ctx, bps, 0, 1, d_ref.data_ptr(), st, d_raw.data_ptr(), 0, d_b.data_ptr(), len(rb)) == 0
        product.svt_amd_synchronize(gpu_ctx)
    bi = np.zeros(nx * ny, BI_BLOCK)
    bi["l0_off"] = bi["l1_off"] = np.arange(nx * ny) * 256
    bi["dst_off"], bi["w"], bi["h"] = by * W + bx, 16, 16
    d_bi = torch.from_numpy(bi.view(np.uint8).copy()).cuda()
    d_dst.zero_()
    torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
    assert product.svt_amd_bipred_clip_batch(gpu_ctx, bps, d_raw0.data_ptr(), d_raw1.data_ptr(), d_dst.data_ptr(), W,
                                             64 + 16384, d_bi.data_ptr(), len(bi)) == 0
    product.svt_amd_synchronize(gpu_ctx)
    pred = d_dst.cpu().numpy().view(dt).reshape(H, W)
    for i in range(0, len(bi), 131):
        r0, r1 = np.zeros(256, np.int16), np.zeros(256, np.int16)
        for rb, out in ((raw_blocks0[i], r0), (raw_blocks1[i], r1)):
            oracle.svt_oracle_mcp(bps, 0, 1, int(rb["fx"]), int(rb["fy"]), flat.ctypes.data + int(rb["ref_off"]) * bps, st,
                                  out.ctypes.data, 0, 16, 16)
        want = np.zeros((16, 16), dt)
        oracle.svt_oracle_BiPredClipping(bps, 16, 16, r0.ctypes.data, r1.ctypes.data, want.ctypes.data, 16, 64 + 16384)
        y, x = divmod(int(bi[i]["dst_off"]), W)
        assert np.array_equal(pred[y:y + 16, x:x + 16], want), i
