"""GPU: the device-resident encode pass svt_amd_encode_lcus (svt-hevc_amd/csrc/encdec_kernels.hip) through the C ABI against
(a) records of the reference's own EncodePass calls (tests/golden/encodepass_*.npz), LCU by LCU in raster order and in the
wavefront batches the host would submit, and (b) the CPU oracle on seeded random coding-unit trees over a larger picture."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_encodepass_golden import CASES, DLF_CASES, INTER_CASES, SAO_CASES, allows_mismatch, compare_lcu, is16, load_case, sao_inputs_of_picture

pytestmark = pytest.mark.gpu


def sig(lib):
    lib.svt_amd_encdec_picture_create.restype = C.c_int
    lib.svt_amd_encdec_picture_create.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_int, C.POINTER(C.c_void_p)]
    for f in (lib.svt_amd_encdec_picture_begin, lib.svt_amd_encdec_picture_destroy):
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    for f in (lib.svt_amd_encode_lcus, lib.svt_amd_encode_lcus16):
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    for f in (lib.svt_amd_encdec_picture_put_borders, lib.svt_amd_encdec_picture_put_borders16):
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]


def encode(lib, ctx, pic, works):
    """8-bit or 16-bit contract by the dtype of the work records"""
    works = np.ascontiguousarray(works)
    wide = works.dtype.itemsize == S.LCU_WORK16_DTYPE.itemsize
    out = np.zeros(len(works), S.LCU_RESULT16_DTYPE if wide else S.LCU_RESULT_DTYPE)
    fn = lib.svt_amd_encode_lcus16 if wide else lib.svt_amd_encode_lcus
    assert fn(ctx, pic, works.ctypes.data, len(works), out.ctypes.data) == 0, lib.svt_amd_last_error()
    return out


def wavefront_batches(wl, hl):
    """LCUs whose left, top and top-right neighbours are done: step s holds the LCUs with x + 2*y == s"""
    for s in range(wl + 2 * (hl - 1)):
        b = [y * wl + (s - 2 * y) for y in range(hl) if 0 <= s - 2 * y < wl]
        if b:
            yield b


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("order", ["raster", "wavefront"])
def test_encode_lcus_matches_reference_records(product, gpu_ctx, name, order):
    lib = product
    sig(lib)
    g, w, h = load_case(name)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    nl = wl * hl
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if is16(g) else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for first in range(0, len(g["work"]), nl):
            set_inter(lib, gpu_ctx, pic, g, {}, first)      # PM-core presets: the picture's rate tables
            assert lib.svt_amd_encdec_picture_begin(gpu_ctx, pic) == 0
            batches = [[i] for i in range(nl)] if order == "raster" else list(wavefront_batches(wl, hl))
            assert sorted(i for b in batches for i in b) == list(range(nl))
            for b in batches:
                idx = [first + i for i in b]
                got = encode(lib, gpu_ctx, pic, g["work"][idx])
                for k, r in zip(idx, got):
                    compare_lcu(g["work"][k], g["result"][k], r, w, h, (name, order, int(g["picture_number"][k]), int(g["lcu_index"][k])))
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def border_of(work, res, w, h):
    """what the host hands over after encoding an LCU itself: last row / column of the un-deblocked LCU and the edge mode types"""
    b = np.zeros(1, S.LCU_BORDER16_DTYPE if res["rec_y"].dtype.itemsize == 2 else S.LCU_BORDER_DTYPE)
    lw, lh = min(64, w - int(work["lcu_x"])), min(64, h - int(work["lcu_y"]))
    b[0]["lcu_x"], b[0]["lcu_y"] = work["lcu_x"], work["lcu_y"]
    b[0]["mode_bottom"][:lw // 4] = 2
    b[0]["mode_right"][:lh // 4] = 2
    ry = res["rec_y"].reshape(64, 64)
    b[0]["bottom_y"][:lw], b[0]["right_y"][:lh] = ry[lh - 1, :lw], ry[:lh, lw - 1]
    for p in ("cb", "cr"):
        rc = res["rec_" + p].reshape(32, 32)
        b[0]["bottom_" + p][:lw // 2], b[0]["right_" + p][:lh // 2] = rc[lh // 2 - 1, :lw // 2], rc[:lh // 2, lw // 2 - 1]
    return b


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("host_lcus", ["odd", "even"])
def test_host_encoded_lcus_enter_the_device_picture(product, gpu_ctx, name, host_lcus):
    """mixed pictures: every second LCU is "encoded by the host" (its recorded reconstruction stands in) and only its last row / column
    and edge mode types are handed to the device; the LCUs the device encodes must still match the records"""
    lib = product
    sig(lib)
    g, w, h = load_case(name)
    put = lib.svt_amd_encdec_picture_put_borders16 if is16(g) else lib.svt_amd_encdec_picture_put_borders
    nl = S.lcu_count(w, h)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if is16(g) else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        set_inter(lib, gpu_ctx, pic, g, {}, 0)      # PM-core presets: the picture's rate tables
        assert lib.svt_amd_encdec_picture_begin(gpu_ctx, pic) == 0
        ndev = 0
        for k in range(nl):
            if (k & 1) == (host_lcus == "odd"):
                b = border_of(g["work"][k], g["result"][k], w, h)
                assert put(gpu_ctx, pic, b.ctypes.data, 1) == 0, lib.svt_amd_last_error()
            else:
                got = encode(lib, gpu_ctx, pic, g["work"][k:k + 1])
                compare_lcu(g["work"][k], g["result"][k], got[0], w, h, (name, host_lcus, k))
                ndev += 1
        assert ndev >= nl // 2
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def random_tree(rng, lw, lh):
    """a random quadtree of 32 / 16 / 8 units over the part of the LCU inside the picture, in Z order"""
    out = []

    def rec(x, y, s):
        if x >= lw or y >= lh:
            return
        if s == 64 or x + s > lw or y + s > lh or (s > 8 and rng.random() < 0.45):
            h = s // 2
            for dy, dx in ((0, 0), (0, h), (h, 0), (h, h)):
                rec(x + dx, y + dy, h)
        else:
            out.append((x, y, s))
    rec(0, 0, 64)
    return out


def z_available(x, y, s):
    """isBottomLeftAvailable / isUpperRightAvailable of a unit inside its LCU (Z-order decoding availability)"""
    def zidx(px, py):
        z = 0
        for b in range(4):
            z |= ((px >> (b + 2)) & 1) << (2 * b) | ((py >> (b + 2)) & 1) << (2 * b + 1)
        return z
    me = zidx(x, y)
    bl = x > 0 and y + s < 64 and zidx(x - 4, y + s) < me
    if x == 0:
        bl = y + s < 64      # the LCU to the left is complete
    tr = y > 0 and x + s < 64 and zidx(x + s, y - 4) < me
    if y == 0:
        tr = True            # the LCU row above is complete (its top-right LCU is a precondition of the call)
    return int(bl), int(tr)


def random_picture(oracle, w, h, qp, seed, tile_cols=1, tile_rows=1, constrained=0):
    """seeded works of a whole picture (random unit trees, modes, QPs, dead zones) and what the CPU oracle makes of them in raster order"""
    oracle.svt_oracle_encode_lcu.restype = None
    oracle.svt_oracle_encode_lcu.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    col0 = [c * wl // tile_cols for c in range(tile_cols)] + [wl]      # EbPictureControlSet.c:743 uniform spacing
    row0 = [r * hl // tile_rows for r in range(tile_rows)] + [hl]
    yy, xx = np.mgrid[0:h, 0:w]
    src = [np.clip(128 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + rng.normal(0, 9, (h, w)), 0, 255).astype(np.uint8)]
    src += [np.clip(128 + 30 * np.sin(xx[::2, ::2] / (31.0 + 9 * k)) + rng.normal(0, 4, (h // 2, w // 2)), 0, 255).astype(np.uint8) for k in range(2)]
    works = np.zeros(wl * hl, S.LCU_WORK_DTYPE)
    for ly in range(hl):
        for lx in range(wl):
            wk = works[ly * wl + lx]
            lw, lh = min(64, w - 64 * lx), min(64, h - 64 * ly)
            wk["lcu_x"], wk["lcu_y"], wk["slice_type"], wk["strong_smoothing"] = 64 * lx, 64 * ly, 2, 1
            wk["constrained_intra"] = 0
            wk["tile_left"], wk["tile_top"], wk["tile_right"] = lx in col0, ly in row0, (lx + 1) in col0
            tree = random_tree(rng, lw, lh)
            wk["num_cus"] = len(tree)
            for i, (x, y, s) in enumerate(tree):
                cu = wk["cu"][i]
                cu["x"], cu["y"], cu["size"], cu["pred_mode"], cu["intra_luma_mode"] = x, y, s, 2, rng.integers(0, 35)
                cu["bottom_left_ok"], cu["top_right_ok"] = z_available(x, y, s)
                cu["qp"] = np.clip(qp + rng.integers(-3, 4), 0, 51)
                cu["chroma_qp"] = min(int(cu["qp"]), 29 + (int(cu["qp"]) - 29) // 2) if cu["qp"] > 29 else cu["qp"]
                cu["dz_offset"] = rng.choice([0, 0, 9, 12])
            sy = np.zeros((64, 64), np.uint8)
            sy[:lh, :lw] = src[0][64 * ly:64 * ly + lh, 64 * lx:64 * lx + lw]
            wk["src_y"] = sy.reshape(-1)
            for p, nm in ((1, "src_cb"), (2, "src_cr")):
                sc = np.zeros((32, 32), np.uint8)
                sc[:lh // 2, :lw // 2] = src[p][32 * ly:32 * ly + lh // 2, 32 * lx:32 * lx + lw // 2]
                wk[nm] = sc.reshape(-1)
    # oracle, raster order
    pitches = (w + 32, w // 2 + 16, w // 2 + 16)
    pb = (C.c_uint32 * 3)(*pitches)
    rec = [np.full((hh, p), 0xA5, np.uint8) for hh, p in zip((h, h // 2, h // 2), pitches)]
    mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
    rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
    want = np.zeros(len(works), S.LCU_RESULT_DTYPE)
    for k in range(len(works)):
        oracle.svt_oracle_encode_lcu(rp, pb, mp.ctypes.data, mp.shape[1], w, h, works[k:k + 1].ctypes.data, want[k:k + 1].ctypes.data)
    return works, want


def random_inter_picture(oracle, w, h, qp, seed, pad=80, intra_lcus=1.0, tile_cols=1):    # pad > 75: a window at a clamped position stays inside the plane
    """a seeded B picture: random unit trees with ~80 % inter units (L0 / L1 / bi, AMVP / merge / skip, whole-LCU 64x64 units, motion vectors
    that reach far outside the picture now and then), two reference pictures, rate tables and per-LCU lambdas spread over three decades -
    and what the CPU oracle makes of it in raster order.  Returns works, want, (reference planes, geometry), cost."""
    from test_oracle_encodepass_golden import inter_oracle_fn
    fn = inter_oracle_fn(oracle, False)
    rng = np.random.default_rng(seed)
    works, _ = random_picture(oracle, w, h, qp, seed, tile_cols)           # trees, intra modes, QPs, source, tile edges
    yy, xx = np.mgrid[0:h + 2 * pad, 0:w + 2 * pad]
    refs = []
    for r in range(2):
        y = np.clip(128 + 60 * np.sin((xx - pad + 3 * r) / 23.0) * np.cos((yy - pad - 2 * r) / 17.0) + rng.normal(0, 6, yy.shape), 0, 255).astype(np.uint8)
        c = [np.clip(128 + 30 * np.sin((xx[::2, ::2] - pad) / (31.0 + 9 * k)) + rng.normal(0, 3, (yy.shape[0] // 2, yy.shape[1] // 2)), 0, 255).astype(np.uint8)
             for k in range(2)]
        refs.append([np.ascontiguousarray(a) for a in (y, c[0], c[1])])
    geom = (w + 2 * pad, (w + 2 * pad) // 2, pad, pad, w, h)
    cost = np.zeros(1, np.dtype([("last", "<u4", 176), ("sig", "u1", 84), ("g1", "u1", 48), ("g2", "u1", 12), ("sigml", "u1", 8), ("g1x", "<u2", 96),
                                 ("sigv", "u1", (32, 16))]))
    assert cost.itemsize == 1560
    cost["last"] = rng.integers(20, 400, 176)
    for k in ("sig", "g1", "g2", "sigml", "sigv"):
        cost[k] = rng.integers(5, 120, cost[k].shape)
    cost["g1x"] = rng.integers(20, 900, 96)
    for wk in works:
        wk["slice_type"], wk["temporal_layer"] = 0, 1
        wk["full_lambda"] = int(10 ** rng.uniform(5.5, 8.4))
        wk["luma_cbf_bits"] = rng.integers(8000, 60000, 4)
        lw, lh = min(64, w - int(wk["lcu_x"])), min(64, h - int(wk["lcu_y"]))
        if lw == 64 and lh == 64 and rng.random() < 0.15:        # one 64x64 unit
            wk["num_cus"] = 1
            cu = wk["cu"][0]
            cu["x"], cu["y"], cu["size"], cu["bottom_left_ok"], cu["top_right_ok"] = 0, 0, 64, 0, 1
        p_inter = 0.8 if rng.random() < intra_lcus else 1.1    # intra_lcus < 1: the other LCUs hold inter units only (they wait for no neighbour)
        for i in range(int(wk["num_cus"])):
            cu = wk["cu"][i]
            if int(cu["size"]) == 64 or rng.random() < p_inter:
                cu["pred_mode"], cu["intra_luma_mode"], cu["dz_offset"] = 1, 0, 0
                cu["inter_dir"], cu["inter_kind"] = rng.choice([0, 1, 2], p=[0.35, 0.25, 0.4]), rng.choice([0, 1, 2], p=[0.4, 0.4, 0.2])
                far = rng.random() < 0.06
                cu["mv"] = rng.integers(-4 * max(w, h), 4 * max(w, h), (2, 2)) if far else rng.integers(-70, 71, (2, 2))
    rs = [S.RefPicture(r[0].ctypes.data, r[1].ctypes.data, r[2].ctypes.data, *geom) for r in refs]
    pitches = (w + 32, w // 2 + 16, w // 2 + 16)
    pb = (C.c_uint32 * 3)(*pitches)
    rec = [np.full((hh, p), 0xA5, np.uint8) for hh, p in zip((h, h // 2, h // 2), pitches)]
    mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
    rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
    want = np.zeros(len(works), S.LCU_RESULT_DTYPE)
    for k in range(len(works)):
        fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data, works[k:k + 1].ctypes.data, want[k:k + 1].ctypes.data)
    return works, want, (refs, geom), cost


def set_inter_random(lib, ctx, pic, refs_geom, cost):
    """the reference pictures of random_inter_picture into HBM + svt_amd_encdec_picture_set_inter; returns what must stay alive"""
    import torch
    refs, geom = refs_geom
    keep = [[torch.from_numpy(a).cuda() for a in r] for r in refs]
    torch.cuda.synchronize()
    rs = [S.RefPicture(k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(), *geom) for k in keep]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
    return keep


@pytest.mark.parametrize("w,h,qp,seed,intra_lcus", [(832, 480, 30, 11, 1.0), (1920, 1080, 34, 12, 1.0), (1920, 1080, 31, 13, 0.12), (3840, 2160, 33, 14, 0.08)])
def test_encode_picture_random_b_pictures_match_oracle(product, oracle, w, h, qp, seed, intra_lcus):
    """full-size B pictures of random inter / intra units against the oracle in raster order: every fractional position of both filters,
    clamped positions far outside the picture, bi-prediction, 64x64 units, the luma cbf decision over three decades of lambda.  intra_lcus < 1:
    intra units in few LCUs only, as in the P / B pictures of an encode - the other LCUs wait for no neighbour and the picture is encoded by as
    many workgroups as the device holds, in no particular order (BASELINE configs[2] size included)"""
    lib = product
    sig_picture(lib)
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    works, want, refs_geom, cost = random_inter_picture(oracle, w, h, qp, seed, intra_lcus=intra_lcus)
    cus = np.concatenate([wk["cu"][:int(wk["num_cus"])] for wk in works])
    amvp = (cus["pred_mode"] == 1) & (cus["inter_kind"] == 0) & (cus["size"] < 64)
    res = np.concatenate([r["cu"][:int(wk["num_cus"])] for wk, r in zip(works, want)])
    dropped = int(((res["nz"][:, 0] != 0) & (res["cbf"][:, 0] == 0) & amvp).sum())
    kept = int(((res["nz"][:, 0] != 0) & (res["cbf"][:, 0] == 1) & amvp).sum())
    assert dropped > 20 and kept > 20, (dropped, kept)          # the decision goes both ways
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        keep = set_inter_random(lib, ctx, pic, refs_geom, cost)
        got = np.zeros(len(works), S.LCU_RESULT_DTYPE)
        assert lib.svt_amd_encode_picture(ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
        for k in range(len(works)):
            compare_lcu(works[k], want[k], got[k], w, h, ("B picture", w, h, k))
        del keep
    finally:
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("w,h,qp,seed", [(832, 480, 27, 1), (456, 264, 38, 2)])
def test_encode_lcus_random_trees_match_oracle(product, oracle, gpu_ctx, w, h, qp, seed):
    lib = product
    sig(lib)
    works, want = random_picture(oracle, w, h, qp, seed)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    # device, wavefront batches
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for rep in range(2):      # the second picture reuses the device picture after _begin
            assert lib.svt_amd_encdec_picture_begin(gpu_ctx, pic) == 0
            for b in wavefront_batches(wl, hl):
                got = encode(lib, gpu_ctx, pic, works[b])
                for k, r in zip(b, got):
                    compare_lcu(works[k], want[k], r, w, h, ("random", rep, k))
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def sig_picture(lib):
    sig(lib)
    for f in (lib.svt_amd_encode_picture, lib.svt_amd_encode_picture16):
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


@pytest.mark.parametrize("name", CASES)
def test_encode_picture_one_call_matches_reference_records(product, gpu_ctx, name):
    """the whole picture in ONE call, wavefront on the device"""
    lib = product
    sig_picture(lib)
    g, w, h = load_case(name)
    nl = S.lcu_count(w, h)
    pic = C.c_void_p()
    wide = is16(g)
    fn = lib.svt_amd_encode_picture16 if wide else lib.svt_amd_encode_picture
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if wide else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for first in range(0, len(g["work"]), nl):
            set_inter(lib, gpu_ctx, pic, g, {}, first)      # PM-core presets: the picture's rate tables
            works = np.ascontiguousarray(g["work"][first:first + nl])
            got = np.zeros(nl, S.LCU_RESULT16_DTYPE if wide else S.LCU_RESULT_DTYPE)
            for rep in range(2):    # twice: the completion flags of the first call must not satisfy the second
                got[:] = 0
                assert fn(gpu_ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
                for k in range(nl):
                    compare_lcu(works[k], g["result"][first + k], got[k], w, h, (name, "picture", rep, k))
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


@pytest.mark.parametrize("w,h,qp,seed,tc,tr", [(1920, 1080, 30, 3, 1, 1), (832, 480, 24, 4, 2, 2), (1280, 768, 35, 5, 4, 1), (3840, 2160, 32, 6, 1, 1)])
def test_encode_picture_random_trees_match_oracle(product, oracle, w, h, qp, seed, tc, tr):
    """full-size pictures (510 / 2040 LCUs >> the resident workgroups) and tile grids: the device wavefront against the oracle in raster order"""
    lib = product
    sig_picture(lib)
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    works, want = random_picture(oracle, w, h, qp, seed, tc, tr)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        got = np.zeros(len(works), S.LCU_RESULT_DTYPE)
        assert lib.svt_amd_encode_picture(ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
        for k in range(len(works)):
            compare_lcu(works[k], want[k], got[k], w, h, ("picture", w, h, k))
    finally:
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_context_destroy(ctx)


def test_encode_lcus_rejects_what_it_does_not_cover(product, gpu_ctx):
    lib = product
    sig(lib)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, 128, 64, 4, C.byref(pic)) != 0      # 1 or 2 bytes per sample
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, 128, 64, 1, C.byref(pic)) == 0
    try:
        wk = np.zeros(1, S.LCU_WORK_DTYPE)
        wk[0]["num_cus"] = 1
        wk[0]["cu"][0]["size"], wk[0]["cu"][0]["pred_mode"] = 32, 1                        # an inter unit without reference pictures
        out = np.zeros(1, S.LCU_RESULT_DTYPE)
        assert lib.svt_amd_encode_lcus(gpu_ctx, pic, wk.ctypes.data, 1, out.ctypes.data) != 0
        wk[0]["cu"][0]["size"], wk[0]["cu"][0]["pred_mode"] = 64, 2                        # a 64x64 unit
        assert lib.svt_amd_encode_lcus(gpu_ctx, pic, wk.ctypes.data, 1, out.ctypes.data) != 0
        wk16, out16 = np.zeros(1, S.LCU_WORK16_DTYPE), np.zeros(1, S.LCU_RESULT16_DTYPE)   # the 16-bit contract on an 8-bit picture
        wk16[0]["num_cus"] = 1
        wk16[0]["cu"][0]["size"], wk16[0]["cu"][0]["pred_mode"] = 32, 2
        assert lib.svt_amd_encode_lcus16(gpu_ctx, pic, wk16.ctypes.data, 1, out16.ctypes.data) != 0
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def device_refs(g, wide):
    """the fixture's reference pictures in HBM: SvtAmdRefPicture records by POC (and the tensors that back them)"""
    import torch
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    keep, out = [], {}
    for i, poc in enumerate(g["ref_pocs"].tolist()):
        planes = [torch.from_numpy(np.ascontiguousarray(g[k][i]).view(np.int16 if wide else np.uint8)).cuda() for k in ("ref_y", "ref_cb", "ref_cr")]
        keep.append(planes)
        out[poc] = S.RefPicture(planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), sy, sc, ox, oy, rw, rh)
    torch.cuda.synchronize()
    return out, keep


def set_inter(lib, ctx, pic, g, refs, k):
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r0, r1 = (refs.get(int(v)) for v in g["ref_poc"][k]) if "ref_poc" in g else (None, None)
    if "cost" not in g or int(g["picture_number"][k]) not in g["cost_pictures"].tolist():
        assert not r0 and not r1
        return            # an I picture without rate tables (only the PM-core presets record them for I pictures)
    cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(int(g["picture_number"][k]))])
    assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None, cost.ctypes.data) == 0, \
        lib.svt_amd_last_error()


@pytest.mark.parametrize("name", INTER_CASES)
@pytest.mark.parametrize("how", ["lcus", "picture"])
def test_p_and_b_pictures_match_reference_records(product, gpu_ctx, name, how):
    """P / B pictures of the reference encoder: inter units (uni / bi-prediction from reference pictures resident in HBM, encode loop,
    luma cbf decision of AMVP units, merge and skip units, 64x64 units) between intra units, LCU by LCU in wavefront batches and as ONE
    launch per picture"""
    lib = product
    sig_picture(lib)
    g, w, h = load_case(name)
    wide = is16(g)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    nl = wl * hl
    refs, keep = device_refs(g, wide)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if wide else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for first in range(0, len(g["work"]), nl):
            set_inter(lib, gpu_ctx, pic, g, refs, first)
            if how == "lcus":
                assert lib.svt_amd_encdec_picture_begin(gpu_ctx, pic) == 0
                got = np.zeros(nl, S.LCU_RESULT16_DTYPE if wide else S.LCU_RESULT_DTYPE)
                for b in wavefront_batches(wl, hl):
                    got[b] = encode(lib, gpu_ctx, pic, g["work"][[first + i for i in b]])
            else:
                works = np.ascontiguousarray(g["work"][first:first + nl])
                got = np.zeros(nl, S.LCU_RESULT16_DTYPE if wide else S.LCU_RESULT_DTYPE)
                fn = lib.svt_amd_encode_picture16 if wide else lib.svt_amd_encode_picture
                assert fn(gpu_ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
            for k in range(nl):
                compare_lcu(g["work"][first + k], g["result"][first + k], got[k], w, h, (name, how, int(g["picture_number"][first]), k),
                            rec=not (int(g["dlf_off"][first + k]) & 2))
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


class DeblockParams(C.Structure):
    """SvtAmdDeblockParams"""
    _fields_ = [("tc_offset", C.c_int8), ("beta_offset", C.c_int8), ("cb_qp_offset", C.c_int8), ("cr_qp_offset", C.c_int8),
                ("slice_type", C.c_uint8), ("pad", C.c_uint8 * 3), ("ref_poc", C.c_uint64 * 2)]


@pytest.mark.parametrize("name", DLF_CASES)
def test_encode_picture_then_deblock_matches_the_encoders_output(product, gpu_ctx, name):
    """two calls per picture - svt_amd_encode_picture (wavefront on the device) and svt_amd_encdec_picture_deblock (boundary strengths
    + deblocking in place on the device picture) - give the reference encoder's own reconstruction output (deblocking on, SAO off),
    8- and 10-bit"""
    lib = product
    sig_picture(lib)
    g, w, h = load_case(name)
    wide = is16(g)
    enc = lib.svt_amd_encode_picture16 if wide else lib.svt_amd_encode_picture
    dbk = lib.svt_amd_encdec_picture_deblock16 if wide else lib.svt_amd_encdec_picture_deblock
    dbk.restype, dbk.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DeblockParams), C.c_void_p, C.c_void_p, C.c_void_p]
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    inter = "ref_pocs" in g          # P / B pictures: prediction from the (deblocked) reference pictures of the fixture, resident in HBM
    refs, keep = device_refs(g, wide) if inter else ({}, None)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if wide else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for first in range(0, len(g["work"]), nl):
            f = int(g["picture_number"][first])      # the encoder's output is in display order
            works = np.ascontiguousarray(g["work"][first:first + nl])
            got = np.zeros(nl, rdt)
            if inter:
                set_inter(lib, gpu_ctx, pic, g, refs, first)
            assert enc(gpu_ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
            for k in range(nl):
                compare_lcu(works[k], g["result"][first + k], got[k], w, h, (name, f, k), rec=False)
            prm = DeblockParams()
            prm.slice_type = int(works[0]["slice_type"])
            if inter:
                prm.ref_poc[0], prm.ref_poc[1] = int(g["ref_poc"][first][0]), int(g["ref_poc"][first][1])
            out = [np.zeros((h, w), sdt), np.zeros((h // 2, w // 2), sdt), np.zeros((h // 2, w // 2), sdt)]
            assert dbk(gpu_ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(prm), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data) == 0, \
                lib.svt_amd_last_error()
            for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                bad = np.argwhere(out[p] != g[nm][f])
                assert len(bad) == 0, (name, f, nm, len(bad), bad[:4].tolist())
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def encoded_picture(lib, ctx, pic, w, h, sdt):
    """the picture object's latest stage, through svt_amd_encdec_picture_reference with the smallest padding, cropped"""
    lib.svt_amd_encdec_picture_reference.restype = C.c_int
    lib.svt_amd_encdec_picture_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ref = S.RefPicture()
    padded = [np.zeros(((h + 16) >> s_, (w + 16) >> s_), sdt) for s_ in (0, 1, 1)]
    assert lib.svt_amd_encdec_picture_reference(ctx, pic, 8, 8, C.byref(ref), *[a.ctypes.data for a in padded]) == 0, lib.svt_amd_last_error()
    return [np.ascontiguousarray(a[(8 >> s_):(8 >> s_) + (h >> s_), (8 >> s_):(8 >> s_) + (w >> s_)]) for a, s_ in zip(padded, (0, 1, 1))]


@pytest.mark.parametrize("name", SAO_CASES)
def test_encode_deblock_sao_on_the_device_matches_the_encoders_output(product, gpu_ctx, name):
    """three calls per picture - svt_amd_encode_picture, svt_amd_encdec_picture_deblock, svt_amd_encdec_picture_sao - and what is in HBM
    (and copied out) is the reference encoder's finished reconstruction with every in-loop filter on; the SAO parameters the device
    decides are the encoder's own, LCU by LCU (I and B pictures, 8- and 10-bit)"""
    from test_oracle_saodec_golden import LCU, same_decision
    lib = product
    sig_picture(lib)
    g, w, h = load_case(name)
    wide = is16(g)
    enc = lib.svt_amd_encode_picture16 if wide else lib.svt_amd_encode_picture
    dbk = lib.svt_amd_encdec_picture_deblock16 if wide else lib.svt_amd_encdec_picture_deblock
    dbk.restype, dbk.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DeblockParams), C.c_void_p, C.c_void_p, C.c_void_p]
    sao = lib.svt_amd_encdec_picture_sao16 if wide else lib.svt_amd_encdec_picture_sao
    sao.restype, sao.argtypes = C.c_int, [C.c_void_p] * 9
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    inter = "ref_pocs" in g
    refs, keep = device_refs(g, wide) if inter else ({}, None)
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if wide else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    decided = 0
    try:
        for first in range(0, len(g["work"]), nl):
            f = int(g["picture_number"][first])
            works = np.ascontiguousarray(g["work"][first:first + nl])
            got = np.zeros(nl, rdt)
            if inter:
                set_inter(lib, gpu_ctx, pic, g, refs, first)
            assert enc(gpu_ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
            prm = DeblockParams()
            prm.slice_type = int(works[0]["slice_type"])
            if inter:
                prm.ref_poc[0], prm.ref_poc[1] = int(g["ref_poc"][first][0]), int(g["ref_poc"][first][1])
            out = [np.zeros((h, w), sdt), np.zeros((h // 2, w // 2), sdt), np.zeros((h // 2, w // 2), sdt)]
            if allows_mismatch(g, w, h, works[0]):   # neither deblocked nor SAO-filtered on the encoder side: the output is the picture as encoded
                P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
                if P is not None:                   # ... but its SAO parameters are decided (on that picture) and signalled
                    lib.svt_amd_encdec_picture_sao_decide.restype, lib.svt_amd_encdec_picture_sao_decide.argtypes = C.c_int, [C.c_void_p] * 6
                    dec = np.zeros(nl, LCU)
                    assert lib.svt_amd_encdec_picture_sao_decide(gpu_ctx, pic, works.ctypes.data, P.ctypes.data, enable.ctypes.data, dec.ctypes.data) == 0, \
                        lib.svt_amd_last_error()
                    for i in idx:
                        assert same_decision(dec[i], want[i]), (name, f, int(i), dec[i], want[i])
                    decided += len(idx)
                out = encoded_picture(lib, gpu_ctx, pic, w, h, sdt)
                for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                    assert np.array_equal(out[p], g[nm][f]), (name, f, nm)
                continue
            assert dbk(gpu_ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(prm), None, None, None) == 0, lib.svt_amd_last_error()
            P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
            if P is None:      # the encode pass shut SAO off for the whole picture: the deblocked picture is the output
                assert dbk(gpu_ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(prm), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data) == 0
            else:
                dec = np.zeros(nl, LCU)
                assert sao(gpu_ctx, pic, works.ctypes.data, P.ctypes.data, enable.ctypes.data, dec.ctypes.data, out[0].ctypes.data, out[1].ctypes.data,
                           out[2].ctypes.data) == 0, lib.svt_amd_last_error()
                for i in idx:
                    assert same_decision(dec[i], want[i]), (name, f, int(i), dec[i], want[i])
                decided += len(idx)
            for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                bad = np.argwhere(out[p] != g[nm][f])
                assert len(bad) == 0, (name, f, nm, len(bad), bad[:4].tolist())
        assert decided >= len(g["work"]) // 2
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def test_picture_sao_needs_the_deblocked_picture(product, gpu_ctx):
    lib = product
    sig_picture(lib)
    lib.svt_amd_encdec_picture_sao.restype, lib.svt_amd_encdec_picture_sao.argtypes = C.c_int, [C.c_void_p] * 9
    pic = C.c_void_p()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, 128, 64, 1, C.byref(pic)) == 0
    try:
        works = np.zeros(2, S.LCU_WORK_DTYPE)
        works[1]["lcu_x"] = 64
        P = np.zeros(88, np.uint8)
        assert lib.svt_amd_encdec_picture_sao(gpu_ctx, pic, works.ctypes.data, P.ctypes.data, None, None, None, None, None) != 0
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


@pytest.mark.parametrize("name", [c for c in SAO_CASES if "_p_" in c or "_b_" in c])
def test_sequence_stays_on_the_device_from_picture_to_picture(product, gpu_ctx, name):
    """a whole sequence (I then P, or random-access B in coding order) with one picture object per picture: encode pass -> deblocking -> SAO ->
    padding, and the result is the NEXT pictures' reference picture through svt_amd_encdec_picture_set_inter - no reconstructed sample comes
    from the host.  Every picture's output equals the encoder's, every padded reference equals the reference picture the encoder used."""
    from test_oracle_saodec_golden import LCU
    lib = product
    sig_picture(lib)
    g, w, h = load_case(name)
    wide = is16(g)
    enc = lib.svt_amd_encode_picture16 if wide else lib.svt_amd_encode_picture
    dbk = lib.svt_amd_encdec_picture_deblock16 if wide else lib.svt_amd_encdec_picture_deblock
    dbk.restype, dbk.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DeblockParams), C.c_void_p, C.c_void_p, C.c_void_p]
    sao = lib.svt_amd_encdec_picture_sao16 if wide else lib.svt_amd_encdec_picture_sao
    sao.restype, sao.argtypes = C.c_int, [C.c_void_p] * 9
    lib.svt_amd_encdec_picture_reference.restype = C.c_int
    lib.svt_amd_encdec_picture_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    firsts = {int(g["picture_number"][k]): k for k in range(0, len(g["work"]), nl)}
    NONE = 0xFFFFFFFFFFFFFFFF
    done, pics, checked_refs = {}, {}, 0
    try:
        while len(done) < len(firsts):
            ready = [f for f, k in firsts.items() if f not in done and all(int(v) == NONE or int(v) in done for v in g["ref_poc"][k])]
            assert ready, (sorted(done), sorted(firsts))
            f = min(ready)
            first = firsts[f]
            works = np.ascontiguousarray(g["work"][first:first + nl])
            pic = C.c_void_p()
            assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 2 if wide else 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
            pics[f] = pic
            r0, r1 = (done.get(int(v)) for v in g["ref_poc"][first])
            if r0 or r1:
                cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)])
                assert lib.svt_amd_encdec_picture_set_inter(gpu_ctx, pic, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None, cost.ctypes.data) == 0, \
                    lib.svt_amd_last_error()
            got = np.zeros(nl, rdt)
            assert enc(gpu_ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
            for k in range(nl):
                compare_lcu(works[k], g["result"][first + k], got[k], w, h, (name, f, k), rec=False)
            prm = DeblockParams()
            prm.slice_type = int(works[0]["slice_type"])
            prm.ref_poc[0], prm.ref_poc[1] = int(g["ref_poc"][first][0]), int(g["ref_poc"][first][1])
            out = [np.zeros((h, w), sdt), np.zeros((h // 2, w // 2), sdt), np.zeros((h // 2, w // 2), sdt)]
            P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
            o = [a.ctypes.data for a in out]
            if allows_mismatch(g, w, h, works[0]):    # the encoder's reference picture is the picture as encoded
                out = encoded_picture(lib, gpu_ctx, pic, w, h, sdt)
            else:
                assert dbk(gpu_ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(prm), *(o if P is None else [None] * 3)) == 0, lib.svt_amd_last_error()
                if P is not None:
                    assert sao(gpu_ctx, pic, works.ctypes.data, P.ctypes.data, enable.ctypes.data, None, *o) == 0, lib.svt_amd_last_error()
            for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                assert np.array_equal(out[p], g[nm][f]), (name, f, nm)
            ref = S.RefPicture()
            padded = [np.zeros(((rh + 2 * oy) >> s_, sy >> s_), sdt) for s_ in (0, 1, 1)]
            assert lib.svt_amd_encdec_picture_reference(gpu_ctx, pic, ox, oy, C.byref(ref), *[a.ctypes.data for a in padded]) == 0, lib.svt_amd_last_error()
            assert (ref.strideY, ref.strideC, ref.originX, ref.originY, ref.width, ref.height) == (sy, sc, ox, oy, rw, rh)
            if f in g["ref_pocs"].tolist():
                i = g["ref_pocs"].tolist().index(f)
                for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):
                    assert np.array_equal(padded[p].reshape(-1), g[nm][i]), (name, f, nm)
                checked_refs += 1
            done[f] = ref
        assert checked_refs == len(g["ref_pocs"])
    finally:
        for pic in pics.values():
            lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


def test_8k_10bit_picture_in_four_tile_columns_encode_and_deblock(product, oracle):
    """BASELINE configs[4] class (7680x4320, 10-bit, encMode 4 = PM-core quantiser, -tile_col_cnt 4) through the picture-level calls: a seeded I picture of
    8,160 LCUs in four tile columns - svt_amd_encode_picture16 against the checker on the first three LCU rows of every tile (the rows below
    depend on them through the intra neighbours, so the whole wavefront is exercised; the checker runs them in raster order) - then
    svt_amd_encdec_picture_deblock16 of the WHOLE picture against the checker's boundary strengths + deblocking of the same un-deblocked samples."""
    from test_oracle_dlf_golden import oracle_bs, oracle_dlf
    from test_oracle_encodepass_golden import deblock_maps
    lib = product
    sig_picture(lib)
    w, h, qp, tiles, rows_checked = 7680, 4320, 34, 4, 3
    rng = np.random.default_rng(41)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    col0 = [c * wl // tiles for c in range(tiles)] + [wl]
    yy, xx = np.mgrid[0:h:8, 0:w:8]
    base = (512 + 240 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.float32)
    src = [np.clip(np.kron(base, np.ones((8, 8), np.float32)) + rng.normal(0, 30, (h, w)).astype(np.float32), 0, 1023).astype(np.uint16)]
    src += [np.clip(512 + rng.normal(0, 20, (h // 2, w // 2)).astype(np.float32), 0, 1023).astype(np.uint16) for _ in range(2)]
    works = np.zeros(wl * hl, S.LCU_WORK16_DTYPE)
    for ly in range(hl):
        for lx in range(wl):
            wk = works[ly * wl + lx]
            lw, lh = min(64, w - 64 * lx), min(64, h - 64 * ly)
            wk["lcu_x"], wk["lcu_y"], wk["slice_type"], wk["strong_smoothing"], wk["pm_core"], wk["full_lambda"] = 64 * lx, 64 * ly, 2, 1, 1, 30000000
            wk["tile_left"], wk["tile_top"], wk["tile_right"] = lx in col0, ly == 0, (lx + 1) in col0
            tree = random_tree(rng, lw, lh)
            wk["num_cus"] = len(tree)
            for i, (x, y, s) in enumerate(tree):
                cu = wk["cu"][i]
                cu["x"], cu["y"], cu["size"], cu["pred_mode"], cu["intra_luma_mode"] = x, y, s, 2, rng.integers(0, 35)
                cu["bottom_left_ok"], cu["top_right_ok"] = z_available(x, y, s)
                cu["qp"] = qp
                cu["chroma_qp"] = 29 + (qp - 29) // 2
            sy = np.zeros((64, 64), np.uint16)
            sy[:lh, :lw] = src[0][64 * ly:64 * ly + lh, 64 * lx:64 * lx + lw]
            wk["src_y"] = sy.reshape(-1)
            for p, nm in ((1, "src_cb"), (2, "src_cr")):
                sc = np.zeros((32, 32), np.uint16)
                sc[:lh // 2, :lw // 2] = src[p][32 * ly:32 * ly + lh // 2, 32 * lx:32 * lx + lw // 2]
                wk[nm] = sc.reshape(-1)
    cost = np.random.default_rng(3).integers(0, 200, 1560, dtype=np.uint8)
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, 640, 384, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()   # the front half's slots are not needed here
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 2, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        lib.svt_amd_encdec_picture_set_inter.restype, lib.svt_amd_encdec_picture_set_inter.argtypes = C.c_int, [C.c_void_p] * 5
        assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, None, None, cost.ctypes.data) == 0, lib.svt_amd_last_error()
        got = np.zeros(len(works), S.LCU_RESULT16_DTYPE)
        assert lib.svt_amd_encode_picture16(ctx, pic, works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
        # the checker on the first rows, raster order, into its own picture
        from test_oracle_encodepass_golden import inter_oracle_fn
        fn = inter_oracle_fn(oracle, True)
        pitches = (w, w // 2, w // 2)
        pb = (C.c_uint32 * 3)(*pitches)
        rec = [np.zeros((h >> s_, p), np.uint16) for s_, p in zip((0, 1, 1), pitches)]
        mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
        rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
        for k in range(rows_checked * wl):
            want = np.zeros(1, S.LCU_RESULT16_DTYPE)
            fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, None, None, cost.ctypes.data, works[k:k + 1].ctypes.data, want.ctypes.data)
            compare_lcu(works[k], want[0], got[k], w, h, ("8k", k))
        assert int(got["cu"]["cbf"].sum()) > 10000
        # deblocking of the whole picture: the device's own un-deblocked samples through the checker
        pre = [np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16)]
        for k, rs in enumerate(got):
            lx, ly = k % wl, k // wl
            lw, lh = min(64, w - 64 * lx), min(64, h - 64 * ly)
            pre[0][64 * ly:64 * ly + lh, 64 * lx:64 * lx + lw] = rs["rec_y"].reshape(64, 64)[:lh, :lw]
            pre[1][32 * ly:32 * ly + lh // 2, 32 * lx:32 * lx + lw // 2] = rs["rec_cb"].reshape(32, 32)[:lh // 2, :lw // 2]
            pre[2][32 * ly:32 * ly + lh // 2, 32 * lx:32 * lx + lw // 2] = rs["rec_cr"].reshape(32, 32)[:lh // 2, :lw // 2]
        cumap, cbf, qpm, edge = deblock_maps(works, got, w, h)
        hdr = dict(width=w, height=h, bytes_per_sample=2, qp_stride=w // 8, tc_offset=0, beta_offset=0, cb_qp_offset=0, cr_qp_offset=0, slice_type=2)
        dp = dict(hdr=hdr, cumap=cumap.reshape(-1), cbf=cbf.reshape(-1), refpoc=np.zeros(2, np.uint64), lcu_edge=edge,
                  bsv=np.zeros((len(works), 256), np.uint8), bsh=np.zeros((len(works), 256), np.uint8))
        dp["bsv"], dp["bsh"] = oracle_bs(oracle, dp)
        dp["pre"], dp["qp"] = pre, qpm.reshape(-1)
        fin = oracle_dlf(oracle, dp)
        dbk = lib.svt_amd_encdec_picture_deblock16
        dbk.restype, dbk.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DeblockParams), C.c_void_p, C.c_void_p, C.c_void_p]
        prm = DeblockParams()
        prm.slice_type = 2
        out = [np.zeros_like(p) for p in pre]
        assert dbk(ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(prm), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data) == 0, lib.svt_amd_last_error()
        for p in range(3):
            bad = np.argwhere(out[p] != fin[p])
            assert len(bad) == 0, ("8k deblock", p, len(bad), bad[:4].tolist())
        assert sum(int((a != b).sum()) for a, b in zip(pre, fin)) > 100000      # the filter did something
    finally:
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_context_destroy(ctx)


def test_4k_b_picture_in_four_tile_columns_by_four_ranks_matches_the_checker(product, oracle):
    """BASELINE configs[2]'s picture size cut the way configs[4] is (tile columns -> ranks): a seeded 4K B picture with four tile columns, each of four logical
    ranks encoding its rectangle with svt_amd_encode_picture_rect - every LCU of every rank against the checker's raster-order encode of the whole picture
    (tiles make the ranks independent: no rank sees another's reconstruction), nothing outside a rank's rectangle touched"""
    from test_recon_exchange import Rect, partition
    lib = product
    sig_picture(lib)
    w, h, world = 3840, 2160, 4
    lib.svt_amd_encode_picture_rect.restype, lib.svt_amd_encode_picture_rect.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Rect)]
    rc, rects, _ = partition(lib, w, h, world, 1, world)
    assert rc == 0
    works, want, refs_geom, cost = random_inter_picture(oracle, w, h, 32, 27, intra_lcus=0.3, tile_cols=world)
    wl = (w + 63) // 64
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, 640, 384, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    pics = []
    try:
        for r in range(world):
            pic = C.c_void_p()
            assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
            pics.append(pic)
            keep = set_inter_random(lib, ctx, pic, refs_geom, cost)
            got = np.zeros(len(works), S.LCU_RESULT_DTYPE)
            assert lib.svt_amd_encode_picture_rect(ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(rects[r])) == 0, lib.svt_amd_last_error()
            x0, x1 = rects[r].x // 64, (rects[r].x + rects[r].w + 63) // 64
            mine = 0
            for k in range(len(works)):
                if x0 <= k % wl < x1:
                    compare_lcu(works[k], want[k], got[k], w, h, ("rank", r, k))
                    mine += 1
                else:
                    assert not got[k]["cu"]["cbf"].any() and not got[k]["rec_y"].any()
            assert mine == (x1 - x0) * ((h + 63) // 64)
            del keep
    finally:
        for pic in pics:
            lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_context_destroy(ctx)
