"""-m gpu: picture-level deblocking (svt_amd_dlf_picture) through the C-ABI against (1) whole pictures of real encoder
runs (tests/golden/dlf_*.npz: reconstruction before / after the reference's per-LCU deblocking drivers) and (2) the
oracle (pinned to the same pictures in tests/test_oracle_dlf_golden.py) on random pictures with random strengths / qps,
8 and 10 bit, strided planes, up to 4K."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_dlf_golden import CASES, load_dlf_case, oracle_dlf

pytestmark = pytest.mark.gpu
vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32


def gpu_dlf(product, gpu_ctx, pic, pad=0):
    """pic as in load_dlf_case; planes are placed in buffers with `pad` extra samples per row -> filtered planes"""
    import torch
    h = pic["hdr"]
    bps = int(h["bytes_per_sample"])
    dev, strides = [], []
    for p in pic["pre"]:
        buf = np.zeros((p.shape[0], p.shape[1] + pad), p.dtype)
        buf[:, :p.shape[1]] = p
        t = torch.from_numpy(buf.view(np.int16) if bps == 2 else buf).cuda()
        dev.append(t)
        strides.append(buf.shape[1])
    aux = [torch.from_numpy(np.ascontiguousarray(pic[k])).cuda() for k in ("bsv", "bsh", "qp")]
    product.svt_amd_dlf_picture.argtypes = [vp, C.c_int, vp, u32, vp, vp, u32, u32, u32, vp, vp, vp, u32, i32, i32, i32, i32]
    torch.cuda.synchronize()
    rc = product.svt_amd_dlf_picture(gpu_ctx, bps, dev[0].data_ptr(), strides[0], dev[1].data_ptr(), dev[2].data_ptr(),
                                     strides[1], int(h["width"]), int(h["height"]), aux[0].data_ptr(), aux[1].data_ptr(),
                                     aux[2].data_ptr(), int(h["qp_stride"]), int(h["tc_offset"]), int(h["beta_offset"]),
                                     int(h["cb_qp_offset"]), int(h["cr_qp_offset"]))
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    out = []
    for t, p in zip(dev, pic["pre"]):
        a = t.cpu().numpy()
        a = a.view(np.uint16) if bps == 2 else a
        out.append(a[:, :p.shape[1]])
    return out


@pytest.mark.parametrize("name", CASES)
def test_dlf_picture_matches_reference_golden(product, gpu_ctx, name):
    for k, pic in enumerate(load_dlf_case(name)):
        got = gpu_dlf(product, gpu_ctx, pic, pad=8 * (k & 1))
        for p in range(3):
            bad = np.argwhere(got[p] != pic["post"][p])
            assert len(bad) == 0, (name, k, p, len(bad), bad[:5].tolist())


def random_picture(rng, w, h, bps, smooth):
    hdr = np.zeros(1, dtype=[("width", "<u4"), ("height", "<u4"), ("bytes_per_sample", "<u4"), ("qp_stride", "<u4"),
                             ("tc_offset", "<i4"), ("beta_offset", "<i4"), ("cb_qp_offset", "<i4"), ("cr_qp_offset", "<i4")])[0]
    hdr["width"], hdr["height"], hdr["bytes_per_sample"], hdr["qp_stride"] = w, h, bps, w // 8 + 3
    hdr["tc_offset"], hdr["beta_offset"] = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
    hdr["cb_qp_offset"], hdr["cr_qp_offset"] = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
    maxv = 255 if bps == 1 else 1023
    planes = []
    for (pw, ph) in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
        if smooth:   # blocky but smooth content: the strong / normal filters actually fire
            base = rng.integers(0, maxv + 1, ((ph + 7) // 8, (pw + 7) // 8))
            a = np.kron(base, np.ones((8, 8), np.int64))[:ph, :pw] + rng.integers(-2, 3, (ph, pw))
        else:
            a = rng.integers(0, maxv + 1, (ph, pw))
        planes.append(np.clip(a, 0, maxv).astype(np.uint8 if bps == 1 else np.uint16))
    nlcu = ((w + 63) // 64) * ((h + 63) // 64)
    bsv, bsh = (rng.integers(0, 3, (nlcu, 256)).astype(np.uint8) for _ in range(2))
    qp = rng.integers(0, 52, (int(hdr["qp_stride"]) * (h // 8),)).astype(np.uint8)
    return dict(hdr=hdr, pre=planes, bsv=bsv, bsh=bsh, qp=qp)


@pytest.mark.parametrize("w,h,bps,smooth", [(64, 64, 1, 1), (8, 8, 1, 1), (72, 40, 1, 0), (416, 240, 2, 1), (1920, 1080, 1, 1),
                                            (3840, 2160, 1, 1), (1280, 720, 2, 0)])
def test_dlf_picture_matches_oracle_random(product, gpu_ctx, oracle, w, h, bps, smooth):
    rng = np.random.default_rng(w * 7 + h + bps)
    pic = random_picture(rng, w, h, bps, smooth)
    want = oracle_dlf(oracle, pic)
    got = gpu_dlf(product, gpu_ctx, pic, pad=4)
    changed = 0
    for p in range(3):
        assert np.array_equal(got[p], want[p]), (p, np.argwhere(got[p] != want[p])[:5].tolist())
        changed += int((want[p] != pic["pre"][p]).sum())
    assert changed > 0 or w <= 8


# ---- picture-level SAO application -------------------------------------------------------------------------------
from test_oracle_dlf_golden import oracle_sao  # noqa: E402


def gpu_sao(product, gpu_ctx, src, bps, width, height, lcus, luma_on, chroma_on, pad=0):
    import torch
    dsrc, ddst, strides = [], [], []
    for p in src:
        buf = np.zeros((p.shape[0], p.shape[1] + pad), p.dtype)
        buf[:, :p.shape[1]] = p
        t = torch.from_numpy(buf.view(np.int16) if bps == 2 else buf).cuda()
        dsrc.append(t)
        ddst.append(torch.zeros_like(t))
        strides.append(buf.shape[1])
    dl = torch.from_numpy(np.ascontiguousarray(lcus).view(np.uint8).copy()).cuda()
    product.svt_amd_sao_apply_picture.argtypes = [vp, C.c_int, vp, vp, u32, u32, u32, u32, vp, C.c_int, C.c_int]
    ps, pd = (vp * 3)(*[t.data_ptr() for t in dsrc]), (vp * 3)(*[t.data_ptr() for t in ddst])
    torch.cuda.synchronize()
    rc = product.svt_amd_sao_apply_picture(gpu_ctx, bps, ps, pd, strides[0], strides[1], width, height, dl.data_ptr(),
                                           int(luma_on), int(chroma_on))
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    out = []
    for t, p in zip(ddst, src):
        a = t.cpu().numpy()
        a = a.view(np.uint16) if bps == 2 else a
        out.append(a[:, :p.shape[1]])
    return out


@pytest.mark.parametrize("name", CASES)
def test_sao_apply_picture_matches_reference_golden(product, gpu_ctx, name):
    for k, pic in enumerate(load_dlf_case(name)):
        h = pic["hdr"]
        got = gpu_sao(product, gpu_ctx, pic["post"], int(h["bytes_per_sample"]), int(h["width"]), int(h["height"]),
                      pic["sao_lcu"], pic["sao_flag"][0], pic["sao_flag"][1], pad=8 * (k & 1))
        for p in range(3):
            bad = np.argwhere(got[p] != pic["final"][p])
            assert len(bad) == 0, (name, k, p, len(bad), bad[:5].tolist())


@pytest.mark.parametrize("w,h,bps,pad", [(64, 64, 1, 0), (72, 40, 1, 3), (424, 240, 1, 0), (416, 240, 2, 8), (1920, 1080, 1, 0),
                                         (3840, 2160, 2, 0), (200, 136, 2, 5)])
def test_sao_apply_picture_matches_oracle_random(product, gpu_ctx, oracle, w, h, bps, pad):
    """random pictures, every type, random tile-edge flags (picture edges always flagged), aligned and unaligned strides"""
    from test_oracle_dlf_golden import load_dlf_case as _l  # noqa: F401
    rng = np.random.default_rng(w + 3 * h + bps)
    maxv = 255 if bps == 1 else 1023
    dt = np.uint8 if bps == 1 else np.uint16
    src = []
    for (pw, ph) in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
        base = rng.integers(0, maxv + 1, ((ph + 3) // 4, (pw + 3) // 4))
        src.append(np.clip(np.kron(base, np.ones((4, 4), np.int64))[:ph, :pw] + rng.integers(-3, 4, (ph, pw)), 0, maxv).astype(dt))
    cols, rows = (w + 63) // 64, (h + 63) // 64
    lcu_dt = np.dtype([("merge_left", "u1"), ("merge_up", "u1"), ("edge_flags", "u1"), ("pad", "u1"), ("type", "<u4", 2),
                       ("offset", "<i4", (3, 4)), ("band", "<u4", 3)])
    lcus = np.zeros(cols * rows, lcu_dt)
    lcus["type"] = rng.integers(0, 6, (cols * rows, 2))
    lcus["offset"] = rng.integers(-7, 8, (cols * rows, 3, 4))
    lcus["band"] = rng.integers(0, 29, (cols * rows, 3))
    ef = rng.integers(0, 16, cols * rows).astype(np.uint8) & rng.integers(0, 16, cols * rows).astype(np.uint8)
    for i in range(cols * rows):
        cx, cy = i % cols, i // cols
        ef[i] |= (1 if cx == 0 else 0) | (2 if cx == cols - 1 else 0) | (4 if cy == 0 else 0) | (8 if cy == rows - 1 else 0)
    lcus["edge_flags"] = ef
    for luma_on, chroma_on in ((1, 1), (1, 0), (0, 1)):
        want = oracle_sao(oracle, src, bps, w, h, lcus, luma_on, chroma_on)
        got = gpu_sao(product, gpu_ctx, src, bps, w, h, lcus, luma_on, chroma_on, pad=pad)
        for p in range(3):
            assert np.array_equal(got[p], want[p]), (p, luma_on, chroma_on, np.argwhere(got[p] != want[p])[:5].tolist())
        if w > 3000:
            break
    assert sum(int((a != b).sum()) for a, b in zip(want, src)) > 0


# ---- picture-level boundary-strength derivation, and the chain maps -> strengths -> deblocked picture ----------------
from test_oracle_dlf_golden import oracle_bs  # noqa: E402


def gpu_bs(product, gpu_ctx, pic):
    import torch
    h = pic["hdr"]
    t = {k: torch.from_numpy(np.ascontiguousarray(pic[k]).view(np.uint8).reshape(-1).copy()).cuda() for k in ("cumap", "cbf", "lcu_edge")}
    d_v = torch.full((pic["bsv"].size,), 9, dtype=torch.uint8, device="cuda")
    d_h = torch.full((pic["bsh"].size,), 9, dtype=torch.uint8, device="cuda")
    product.svt_amd_bs_picture.argtypes = [vp, vp, vp, u32, u32, C.c_int, C.c_uint64, C.c_uint64, vp, vp, vp]
    torch.cuda.synchronize()
    rc = product.svt_amd_bs_picture(gpu_ctx, t["cumap"].data_ptr(), t["cbf"].data_ptr(), int(h["width"]), int(h["height"]),
                                    int(h["slice_type"]), int(pic["refpoc"][0]), int(pic["refpoc"][1]), t["lcu_edge"].data_ptr(),
                                    d_v.data_ptr(), d_h.data_ptr())
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    return d_v.cpu().numpy().reshape(pic["bsv"].shape), d_h.cpu().numpy().reshape(pic["bsh"].shape)


@pytest.mark.parametrize("name", CASES)
def test_bs_picture_matches_reference_golden(product, gpu_ctx, name):
    for k, pic in enumerate(load_dlf_case(name)):
        bsv, bsh = gpu_bs(product, gpu_ctx, pic)
        assert np.array_equal(bsv, pic["bsv"]) and np.array_equal(bsh, pic["bsh"]), (name, k)
        # chained on the device results: strengths -> deblocked picture
        got = gpu_dlf(product, gpu_ctx, dict(pic, bsv=bsv, bsh=bsh))
        for p in range(3):
            assert np.array_equal(got[p], pic["post"][p]), (name, k, p)


def test_bs_picture_matches_oracle_random(product, gpu_ctx, oracle):
    """random coding-unit trees, modes, vectors and cbf maps on B and P slices with equal and different reference POCs"""
    rng = np.random.default_rng(8)
    W, H = 448, 200
    bw, bh = W // 8, H // 8
    cu_dt = np.dtype([("mode", "u1"), ("dir", "u1"), ("size_log2", "u1"), ("pad", "u1"), ("mv", "<i2", (2, 2))])
    for trial in range(6):
        cumap = np.zeros((bh, bw), cu_dt)
        for ly in range(0, bh, 8):
            for lx in range(0, bw, 8):                       # one LCU: random quadtree
                def fill(x0, y0, lg):
                    n = 1 << (lg - 3)
                    if lg > 3 and rng.random() < 0.55:
                        for dy in (0, n // 2):
                            for dx in (0, n // 2):
                                fill(x0 + dx, y0 + dy, lg - 1)
                        return
                    e = (int(rng.choice([1, 1, 2])), int(rng.integers(0, 3)), lg, 0, rng.integers(-9, 10, (2, 2)) * int(rng.choice([1, 1, 3])))
                    cumap[y0:y0 + n, x0:x0 + n] = np.array([e], cu_dt)[0]
                fill(lx, ly, 6)
        cumap = cumap[:bh, :bw]
        cbf = (rng.random((H // 4, W // 4)) < 0.3).astype(np.uint8)
        nlcu = ((W + 63) // 64) * ((H + 63) // 64)
        edge = np.zeros(nlcu, np.uint8)
        cols = (W + 63) // 64
        for i in range(nlcu):
            edge[i] = (1 if i % cols == 0 or rng.random() < 0.2 else 0) | (2 if i // cols == 0 or rng.random() < 0.2 else 0)
        hdr = np.zeros(1, dtype=[("width", "<u4"), ("height", "<u4"), ("slice_type", "<u4")])[0]
        hdr["width"], hdr["height"], hdr["slice_type"] = W, H, trial % 2
        poc = np.array([5, 5 if trial < 3 else 9], np.uint64)
        pic = dict(hdr=hdr, cumap=cumap.reshape(-1), cbf=cbf.reshape(-1), refpoc=poc, lcu_edge=edge, bsv=np.zeros((nlcu, 256), np.uint8),
                   bsh=np.zeros((nlcu, 256), np.uint8))
        want = oracle_bs(oracle, pic)
        got = gpu_bs(product, gpu_ctx, pic)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), trial
        assert set(np.unique(want[0]).tolist()) == {0, 1, 2}
