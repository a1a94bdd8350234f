"""Entropy hand-off pre-scan on the device (svt-hevc_amd/csrc/coeffscan_kernels.hip) through the C-ABI: svt_amd_coeff_scan_picture against the
CPU checker (pinned on the reference's coder by tests/test_oracle_coeffscan.py) - block records, sub-block groups and levels bit for bit - on the
recorded encodes, on a seeded 4K picture from device arrays, and, where oracle/_ref travels with the snapshot, straight against the reference's coder."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_encodepass_golden import ALL, is16, load_case

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
CASES = [c for c in ALL if not c.startswith(("dlf_", "sao_"))]


def sig(lib):
    lib.svt_amd_coeff_scan_picture.restype = C.c_int
    lib.svt_amd_coeff_scan_picture.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, u32, vp, u32, C.POINTER(u32 * 2)]


def device_scan(lib, ctx, works, wstride, results, rstride, device, n, gcap=None, lcap=None):
    gcap, lcap = gcap if gcap is not None else 384 * n, lcap if lcap is not None else 6144 * n
    lcus, groups, levels = np.zeros(n, S.COEFF_SCAN_LCU_DTYPE), np.zeros(max(gcap, 1), S.COEFF_SCAN_GROUP_DTYPE), np.zeros(max(lcap, 1), np.uint16)
    totals = (u32 * 2)()
    rc = lib.svt_amd_coeff_scan_picture(ctx, works, wstride, results, rstride, device, n, lcus.ctypes.data, groups.ctypes.data, gcap, levels.ctypes.data, lcap,
                                        C.byref(totals))
    return rc, lcus, groups, levels, (int(totals[0]), int(totals[1]))


def oracle_scan(oracle, work, res):
    oracle.svt_oracle_coeff_scan_lcu.restype, oracle.svt_oracle_coeff_scan_lcu.argtypes = C.c_int, [vp, vp, vp, vp, vp]
    lcu = np.zeros(1, S.COEFF_SCAN_LCU_DTYPE)
    groups, levels = np.zeros(384, S.COEFF_SCAN_GROUP_DTYPE), np.zeros(6144, np.uint16)
    oracle.svt_oracle_coeff_scan_lcu(work.ctypes.data, res.ctypes.data, lcu.ctypes.data, groups.ctypes.data, levels.ctypes.data)
    return lcu[0], groups, levels


def compare_lcu(tag, want, wg, wl, got, groups, levels):
    assert int(got["groups"]) == int(want["groups"]) and int(got["levels"]) == int(want["levels"]), (tag, got["groups"], want["groups"], got["levels"], want["levels"])
    assert np.array_equal(got["tu"], want["tu"]), (tag, np.argwhere(got["tu"] != want["tu"])[:4].tolist())
    gb, lb, ng, nl = int(got["group_base"]), int(got["level_base"]), int(got["groups"]), int(got["levels"])
    assert np.array_equal(groups[gb:gb + ng], wg[:ng]), (tag, "groups")
    assert np.array_equal(levels[lb:lb + nl], wl[:nl]), (tag, "levels")


@pytest.mark.parametrize("name", CASES)
def test_prescan_of_recorded_pictures_matches_the_checker(product, oracle, gpu_ctx, name):
    lib = product
    sig(lib)
    g, w, h = load_case(name)
    works, results = np.ascontiguousarray(g["work"]), np.ascontiguousarray(g["result"])
    n = len(works)
    rc, lcus, groups, levels, totals = device_scan(lib, gpu_ctx, works.ctypes.data, works.dtype.itemsize, results.ctypes.data, results.dtype.itemsize, 0, n)
    assert rc == 0, lib.svt_amd_last_error()
    gsum = lsum = 0
    for k in range(n):
        want, wg, wl = oracle_scan(oracle, works[k:k + 1], results[k:k + 1])
        assert int(lcus[k]["group_base"]) == gsum and int(lcus[k]["level_base"]) == lsum, (name, k)
        compare_lcu((name, k), want, wg, wl, lcus[k], groups, levels)
        gsum, lsum = gsum + int(want["groups"]), lsum + int(want["levels"])
    assert totals == (gsum, lsum) and lsum > 0


def random_records(w, h, seed, density):
    """unit trees of tools/encodepass_bench.py with random coefficient planes: cbf / count records consistent with the planes"""
    import sys, os
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import encodepass_bench as EPB
    rng = np.random.default_rng(seed)
    works = EPB.works_of(w, h, 30, seed, None, 0.5)
    n = len(works)
    results = np.zeros(n, S.LCU_RESULT_DTYPE)
    yy, xx = np.mgrid[:64, :64]
    for k in range(n):
        wk, rs = works[k], results[k]
        for c in range(int(wk["num_cus"])):
            cu = wk["cu"][c]
            x, y, s = int(cu["x"]), int(cu["y"]), int(cu["size"])
            for p in range(3):
                ts = s if p == 0 else (4 if s == 8 else s // 2)
                if rng.random() < 0.25:
                    continue                                        # cbf 0: whatever the plane holds there is not coded
                kind = rng.random()
                if kind < 0.2:
                    blk = np.zeros((ts, ts), np.int16)
                    blk[0, 0] = rng.choice([-2, 1, 4])
                else:
                    fy, fx = np.mgrid[:ts, :ts]
                    blk = (rng.integers(-60, 61, (ts, ts)) * (rng.random((ts, ts)) < density) / (1 + (fx + fy) ** 1.3)).astype(np.int16)
                if not blk.any():
                    continue
                plane = rs["coeff_y"].reshape(64, 64) if p == 0 else rs[("coeff_cb", "coeff_cr")[p - 1]].reshape(32, 32)
                ox, oy = (x, y) if p == 0 else (x // 2, y // 2)
                plane[oy:oy + ts, ox:ox + ts] = blk
                rs["cu"][c]["cbf"][p], rs["cu"][c]["nz"][p] = 1, int((blk != 0).sum())
    return works, results


@pytest.mark.parametrize("w,h,seed,density", [(3840, 2160, 21, 0.5), (1920, 1080, 22, 0.9)])
def test_prescan_of_a_seeded_picture_from_device_arrays(product, oracle, w, h, seed, density):
    """BASELINE configs[2] / [1] picture sizes: records resident in HBM (what svt_amd_encode_picture_device leaves); every 7th LCU against the checker,
    all of them through what does not depend on it: the lists are packed in LCU order and hold exactly the non-zero coefficients of the coded blocks"""
    import torch
    lib = product
    sig(lib)
    ctx = vp()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        works, results = random_records(w, h, seed, density)
        n = len(works)
        dw, dr = torch.from_numpy(works.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(results.view(np.uint8).reshape(-1)).cuda()
        torch.cuda.synchronize()
        rc, lcus, groups, levels, totals = device_scan(lib, ctx, dw.data_ptr(), works.dtype.itemsize, dr.data_ptr(), results.dtype.itemsize, 1, n)
        assert rc == 0, lib.svt_amd_last_error()
        assert np.array_equal(lcus["group_base"], np.concatenate([[0], np.cumsum(lcus["groups"].astype(np.int64))[:-1]]))
        assert np.array_equal(lcus["level_base"], np.concatenate([[0], np.cumsum(lcus["levels"].astype(np.int64))[:-1]]))
        assert totals == (int(lcus["groups"].sum()), int(lcus["levels"].sum()))
        coded = int(sum(int(results[k]["cu"]["nz"][:int(works[k]["num_cus"])][results[k]["cu"]["cbf"][:int(works[k]["num_cus"])] != 0].sum()) for k in range(n)))
        assert totals[1] == coded and (levels[:totals[1]] != 0).all()
        for k in range(0, n, 7):
            want, wg, wl = oracle_scan(oracle, works[k:k + 1], results[k:k + 1])
            compare_lcu((w, h, k), want, wg, wl, lcus[k], groups, levels)
        # the capacities are checked, not trusted
        rc, _, _, _, need = device_scan(lib, ctx, dw.data_ptr(), works.dtype.itemsize, dr.data_ptr(), results.dtype.itemsize, 1, n, gcap=totals[0] - 1, lcap=totals[1])
        assert rc != 0 and need == totals
    finally:
        lib.svt_amd_context_destroy(ctx)


def test_device_records_code_the_reference_bytes(product, gpu_ctx):
    """no checker in between: the device's records through the CABAC loop of integration/svt_coeff_scan_consumer.h, the recorded coefficients through the
    reference's EncodeQuantizedCoefficients_generic, on two states of the reference's arithmetic coder (needs oracle/_ref on the box)"""
    ref = S.load_ref()
    if ref is None or not hasattr(ref, "svt_ref_cabac_new"):
        pytest.skip("oracle/_ref/libsvtref.so is not on this box")
    from test_oracle_coeffscan import state_of
    ref.svt_ref_cabac_new.restype, ref.svt_ref_cabac_new.argtypes = vp, [u32]
    ref.svt_ref_cabac_free.restype, ref.svt_ref_cabac_free.argtypes = None, [vp]
    ref.svt_ref_cabac_code_raw.restype, ref.svt_ref_cabac_code_raw.argtypes = None, [vp, u32, u32, u32, vp, u32, u32, u32, C.c_int]
    ref.svt_ref_cabac_code_scan.restype, ref.svt_ref_cabac_code_scan.argtypes = None, [vp, u32, u32, vp, vp, vp]
    ref.svt_ref_cabac_state.restype, ref.svt_ref_cabac_state.argtypes = u32, [vp, vp, u32]
    lib = product
    sig(lib)
    for name in ("b_motion_320x192_m5", "i_motion_416x240_m9", "p10_motion_320x192_m7"):
        g, w, h = load_case(name)
        works, results = np.ascontiguousarray(g["work"]), np.ascontiguousarray(g["result"])
        n = len(works)
        rc, lcus, groups, levels, totals = device_scan(lib, gpu_ctx, works.ctypes.data, works.dtype.itemsize, results.ctypes.data, results.dtype.itemsize, 0, n)
        assert rc == 0, lib.svt_amd_last_error()
        a, b = ref.svt_ref_cabac_new(7), ref.svt_ref_cabac_new(7)
        try:
            blocks = 0
            for k in range(n):
                wk, rs = works[k], results[k]
                ncu = int(wk["num_cus"])
                big = ncu == 1 and int(wk["cu"][0]["size"]) == 64
                gl = np.ascontiguousarray(groups[int(lcus[k]["group_base"]):int(lcus[k]["group_base"]) + 384])
                ll = np.ascontiguousarray(levels[int(lcus[k]["level_base"]):int(lcus[k]["level_base"]) + 6144])
                for c in (range(1, 5) if big else range(ncu)):
                    cu = wk["cu"][0 if big else c]
                    size = 32 if big else int(cu["size"])
                    x, y = (32 * ((c - 1) & 1), 32 * ((c - 1) >> 1)) if big else (int(cu["x"]), int(cu["y"]))
                    for p in range(3):
                        if not rs["cu"][c]["cbf"][p]:
                            continue
                        ts = size if p == 0 else (4 if size == 8 else size // 2)
                        plane = rs["coeff_y"].reshape(64, 64) if p == 0 else rs[("coeff_cb", "coeff_cr")[p - 1]].reshape(32, 32)
                        st = 64 if p == 0 else 32
                        full = np.zeros((32, st), np.int16)
                        full[:ts, :ts] = plane[(y >> (p > 0)):, (x >> (p > 0)):][:ts, :ts]
                        ref.svt_ref_cabac_code_raw(a, ts, int(cu["pred_mode"]), int(cu["intra_luma_mode"]), full.ctypes.data, st, p, int(rs["cu"][c]["nz"][p]), 0)
                        tu = np.ascontiguousarray(lcus[k]["tu"][p][c:c + 1])
                        ref.svt_ref_cabac_code_scan(b, ts, p, tu.ctypes.data, gl.ctypes.data, ll.ctypes.data)
                        blocks += 1
            sa, sb = state_of(ref, a), state_of(ref, b)
            assert blocks > 20 and np.array_equal(sa, sb), (name, blocks)
        finally:
            ref.svt_ref_cabac_free(a), ref.svt_ref_cabac_free(b)
