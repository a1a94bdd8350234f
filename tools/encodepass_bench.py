#!/usr/bin/env python3
"""Device-resident encode pass at picture level (needs the GPU): svt_amd_encode_picture_device - ONE launch per picture, wavefront on
the device, work / result arrays in HBM - on seeded all-intra pictures with random unit trees (tests/test_gpu_encodepass.py's
generator).  Reports ms per picture, LCUs/s and pictures/s for 1..P pictures in flight on as many lanes.
usage: python tools/encodepass_bench.py [width height] [iters]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402
from test_gpu_encodepass import random_tree, z_available  # noqa: E402


def works_of(w, h, qp, seed, only=None, inter=0.0, intra_lcus=1.0):
    """intra_lcus: fraction of the LCUs that may hold intra units (the others are inter units only, as most LCUs of a real P / B picture)"""
    rng = np.random.default_rng(seed)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    works = np.zeros(wl * hl, S.LCU_WORK_DTYPE)
    src = rng.integers(0, 256, (3, 64, 64), dtype=np.uint8)
    for ly in range(hl):
        for lx in range(wl):
            wk = works[ly * wl + lx]
            lw, lh = min(64, w - 64 * lx), min(64, h - 64 * ly)
            wk["lcu_x"], wk["lcu_y"], wk["slice_type"], wk["strong_smoothing"] = 64 * lx, 64 * ly, 2, 1
            wk["tile_left"], wk["tile_top"], wk["tile_right"] = lx == 0, ly == 0, lx == wl - 1
            if only is None:
                tree = random_tree(rng, lw, lh)
            else:  # units of `only` where they fit, 8x8 units in what is left of a partial LCU
                tree = [(x, y, only) for y in range(0, lh - only + 1, only) for x in range(0, lw - only + 1, only)]
                tree += [(x, y, 8) for y in range(0, lh, 8) for x in range(0, lw, 8) if x >= lw // only * only or y >= lh // only * only]
            if only is not None:   # Z order inside the LCU
                tree.sort(key=lambda u: sum((((u[0] >> (3 + b)) & 1) << (2 * b)) | (((u[1] >> (3 + b)) & 1) << (2 * b + 1)) for b in range(3)))
            wk["num_cus"] = len(tree)
            inter_lcu = inter if rng.random() < intra_lcus else 1.1
            for i, (x, y, s) in enumerate(tree):
                cu = wk["cu"][i]
                cu["x"], cu["y"], cu["size"], cu["pred_mode"], cu["intra_luma_mode"] = x, y, s, 2, rng.integers(0, 35)
                cu["bottom_left_ok"], cu["top_right_ok"] = z_available(x, y, s)
                cu["qp"], cu["chroma_qp"] = qp, min(qp, 29 + (qp - 29) // 2) if qp > 29 else qp
                if rng.random() < inter_lcu:   # an inter unit: B-picture mix of directions and kinds, motion within +-24 samples (quarter units)
                    cu["pred_mode"], cu["intra_luma_mode"] = 1, 0
                    cu["inter_dir"], cu["inter_kind"] = rng.choice([0, 1, 2], p=[0.4, 0.2, 0.4]), rng.choice([0, 1, 2], p=[0.15, 0.55, 0.3])
                    cu["mv"] = rng.integers(-96, 97, (2, 2))
            if inter:
                wk["slice_type"], wk["full_lambda"], wk["luma_cbf_bits"] = 0, 60000000, (20000, 28000, 50000, 38000)
            wk["src_y"] = np.roll(src[0], (lx, ly), (0, 1)).reshape(-1)
            wk["src_cb"], wk["src_cr"] = src[1, :32, :32].reshape(-1), src[2, :32, :32].reshape(-1)
    return works


def setup(lib):
    vp = C.c_void_p
    lib.svt_amd_encdec_picture_create.argtypes = [vp, C.c_uint16, C.c_uint16, C.c_int, C.POINTER(vp)]
    lib.svt_amd_encode_picture_device.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.svt_amd_encode_picture_device_inter.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
    lib.svt_amd_context_fork.argtypes = [vp, C.POINTER(vp)]
    lib.svt_amd_debug_encdec_profile.argtypes = [vp, vp, vp]
    lib.svt_amd_encdec_picture_set_inter.argtypes = [vp, vp, vp, vp, vp]


def b_picture_inputs(W, H, pad=80):
    """a B picture's picture-level inputs: two reference pictures resident in HBM (padded planes) and coefficient-rate tables"""
    gen = torch.Generator(device="cuda").manual_seed(5)
    planes = [[torch.randint(0, 256, ((H + 2 * pad) >> sh, (W + 2 * pad) >> sh), dtype=torch.uint8, device="cuda", generator=gen) for sh in (0, 1, 1)]
              for _ in range(2)]
    refs = [S.RefPicture(pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), W + 2 * pad, (W + 2 * pad) >> 1, pad, pad, W, H) for pl in planes]
    cost = np.random.default_rng(3).integers(0, 200, 1560, dtype=np.uint8)
    return planes, refs, cost


def run_content(lib, root, W, H, works, flights, iters, inter_inputs=None, profile=False):
    """`flights` pictures in flight on as many lanes, `iters` rounds: seconds per round (+ the per-LCU clock sums of one picture)"""
    vp = C.c_void_p
    nl = S.lcu_count(W, H)
    lanes, pics, dws, drs = [], [], [], []
    for i in range(flights):
        lane, pic = vp(), vp()
        assert lib.svt_amd_context_fork(root, C.byref(lane)) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_encdec_picture_create(lane, W, H, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        if inter_inputs:
            _, refs, cost = inter_inputs
            assert lib.svt_amd_encdec_picture_set_inter(lane, pic, C.byref(refs[0]), C.byref(refs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
        dws.append(torch.from_numpy(works.view(np.uint8).reshape(-1)).cuda())
        drs.append(torch.empty(nl * S.LCU_RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda"))
        lanes.append(lane), pics.append(pic)
    torch.cuda.synchronize()

    free = int(sum(1 for wk in works if not (wk["cu"]["pred_mode"][:int(wk["num_cus"])] == 2).any()))

    def go():
        for i in range(flights):
            assert lib.svt_amd_encode_picture_device_inter(lanes[i], pics[i], dws[i].data_ptr(), drs[i].data_ptr(), 1, free) == 0, lib.svt_amd_last_error()
    go()
    for lane in lanes:
        lib.svt_amd_synchronize(lane)
    t0 = time.perf_counter()
    for _ in range(iters):
        go()
    for lane in lanes:
        lib.svt_amd_synchronize(lane)
    dt = (time.perf_counter() - t0) / iters
    prof = None
    if profile:   # where an LCU's time goes (shader clocks of thread 0)
        assert lib.svt_amd_debug_encdec_profile(lanes[0], pics[0], None) == 0
        go()
        prof = np.zeros((nl, 16), np.uint64)
        assert lib.svt_amd_debug_encdec_profile(lanes[0], pics[0], prof.ctypes.data) == 0
    for lane, pic in zip(lanes, pics):
        lib.svt_amd_encdec_picture_destroy(lane, pic)
        lib.svt_amd_context_destroy(lane)
    return dt, prof


def finish_chain_ms(lib, root, W, H, works, inputs, iters=3):
    """the whole closed-loop tail of one B picture through the host-array ABI: svt_amd_encode_picture (work records up, results down) ->
    svt_amd_encdec_picture_deblock -> svt_amd_encdec_picture_sao -> svt_amd_encdec_picture_reference; ms per stage (host clock, blocking calls)"""
    vp = C.c_void_p

    class DeblockParams(C.Structure):
        _fields_ = [("tc_offset", C.c_int8), ("beta_offset", C.c_int8), ("cb_qp_offset", C.c_int8), ("cr_qp_offset", C.c_int8),
                    ("slice_type", C.c_uint8), ("pad", C.c_uint8 * 3), ("ref_poc", C.c_uint64 * 2)]
    lib.svt_amd_encode_picture.argtypes = [vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_deblock.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_sao.argtypes = [vp] * 9
    lib.svt_amd_encdec_picture_reference.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    nl = S.lcu_count(W, H)
    lane, pic = vp(), vp()
    assert lib.svt_amd_context_fork(root, C.byref(lane)) == 0
    assert lib.svt_amd_encdec_picture_create(lane, W, H, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    _, refs, cost = inputs
    assert lib.svt_amd_encdec_picture_set_inter(lane, pic, C.byref(refs[0]), C.byref(refs[1]), cost.ctypes.data) == 0
    res = np.zeros(nl, S.LCU_RESULT_DTYPE)
    prm = DeblockParams()
    prm.slice_type, prm.ref_poc[0], prm.ref_poc[1] = 0, 0, 8
    P = np.zeros(88, np.uint8)       # SvtAmdSaoDecisionParams: lambdas, rate tables, mm_sao
    P.view("<u8")[0:2] = (60000000, 50000000)
    P.view("<u4")[4:20] = np.arange(3000, 3000 + 16 * 700, 700)
    P[81] = 1                        # mm_sao
    ref = S.RefPicture()
    t = {"encode": 0.0, "deblock": 0.0, "sao": 0.0, "pad": 0.0}
    for it in range(iters + 1):
        t0 = time.perf_counter()
        assert lib.svt_amd_encode_picture(lane, pic, works.ctypes.data, res.ctypes.data) == 0, lib.svt_amd_last_error()
        t1 = time.perf_counter()
        assert lib.svt_amd_encdec_picture_deblock(lane, pic, works.ctypes.data, res.ctypes.data, C.byref(prm), None, None, None) == 0, lib.svt_amd_last_error()
        t2 = time.perf_counter()
        assert lib.svt_amd_encdec_picture_sao(lane, pic, works.ctypes.data, P.ctypes.data, None, None, None, None, None) == 0, lib.svt_amd_last_error()
        t3 = time.perf_counter()
        assert lib.svt_amd_encdec_picture_reference(lane, pic, 80, 80, C.byref(ref), None, None, None) == 0, lib.svt_amd_last_error()
        t4 = time.perf_counter()
        if it:   # the first round allocates
            for k, v in (("encode", t1 - t0), ("deblock", t2 - t1), ("sao", t3 - t2), ("pad", t4 - t3)):
                t[k] += v
    lib.svt_amd_encdec_picture_destroy(lane, pic)
    lib.svt_amd_context_destroy(lane)
    out = {k + "_ms": round(v / iters * 1e3, 2) for k, v in t.items()}
    out["total_ms"] = round(sum(t.values()) / iters * 1e3, 2)
    out["what"] = "one 4K B picture from work records on the host to a padded reference picture in HBM: encode pass (records over PCIe both ways), deblocking, SAO, padding"
    return out


def tile_column_works(W, H, cols):
    """the B picture of tile_ranks_leg: `cols` tile columns on the reference's uniform tile grid (EbPictureControlSet.c:743); the same on every rank (seeded)"""
    works = works_of(W, H, 32, 9, None, 0.85, 0.1)
    wl = (W + 63) // 64
    starts = {c * wl // cols for c in range(cols)}
    for wk in works:
        lx = int(wk["lcu_x"]) // 64
        wk["tile_left"], wk["tile_right"] = lx in starts, (lx + 1 in starts) or lx == wl - 1
    return works


def tile_ranks_leg(lib, ctx, rank, world, barrier=None, W=3840, H=2160, iters=5):
    """bench.py --gpus N: ONE 4K B picture cut into `world` tile columns, rank r encoding its rectangle (svt_amd_encode_picture_rect, host-array ABI:
    every rank uploads the unit lists, downloads its results), then the finished planes all-gathered (svt_amd_encdec_picture_exchange; needs the
    context's communicator at world > 1) and padded into a reference picture.  Seconds per picture on this rank (the caller takes the max)."""
    vp = C.c_void_p
    setup(lib)

    class Rect(C.Structure):
        _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint16), ("h", C.c_uint16)]
    lib.svt_amd_tile_partition.argtypes = [C.c_uint16, C.c_uint16, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(C.c_int)]
    lib.svt_amd_encode_picture_rect.argtypes = [vp, vp, vp, vp, C.POINTER(Rect)]
    lib.svt_amd_encdec_picture_exchange.argtypes = [vp, vp, C.POINTER(Rect), C.c_int, C.c_int]
    lib.svt_amd_encdec_picture_reference.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    rects = (Rect * world)()
    assert lib.svt_amd_tile_partition(W, H, world, 1, world, rects, None) == 0, lib.svt_amd_last_error()
    works = tile_column_works(W, H, world)
    inputs = b_picture_inputs(W, H)
    nl = S.lcu_count(W, H)
    pic = vp()
    assert lib.svt_amd_encdec_picture_create(ctx, W, H, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    _, refs, cost = inputs
    assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, C.byref(refs[0]), C.byref(refs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
    res = np.zeros(nl, S.LCU_RESULT_DTYPE)
    ref = S.RefPicture()
    t = {"encode": 0.0, "exchange": 0.0, "pad": 0.0}
    for it in range(iters + 1):
        if barrier:
            barrier()
        t0 = time.perf_counter()
        assert lib.svt_amd_encode_picture_rect(ctx, pic, works.ctypes.data, res.ctypes.data, C.byref(rects[rank])) == 0, lib.svt_amd_last_error()
        t1 = time.perf_counter()
        assert lib.svt_amd_encdec_picture_exchange(ctx, pic, rects, world, rank) == 0, lib.svt_amd_last_error()
        lib.svt_amd_synchronize(ctx)
        t2 = time.perf_counter()
        assert lib.svt_amd_encdec_picture_reference(ctx, pic, 80, 80, C.byref(ref), None, None, None) == 0, lib.svt_amd_last_error()
        t3 = time.perf_counter()
        if it:
            t["encode"] += t1 - t0
            t["exchange"] += t2 - t1
            t["pad"] += t3 - t2
    lib.svt_amd_encdec_picture_destroy(ctx, pic)
    mine = sum(1 for wk in works if rects[rank].x <= int(wk["lcu_x"]) < rects[rank].x + rects[rank].w)
    return {"picture": "%dx%d B picture, %d tile columns over %d ranks" % (W, H, world, world), "lcus_of_this_rank": mine, "lcus": nl,
            "encode_ms": round(t["encode"] / iters * 1e3, 2), "exchange_ms": round(t["exchange"] / iters * 1e3, 2), "pad_ms": round(t["pad"] / iters * 1e3, 2),
            "seconds_per_picture": sum(t.values()) / iters}


def measure_b_picture(lib, root, W=3840, H=2160, iters=5):
    """bench.py's `encode_pass` leg: a 4K B picture of random unit trees, 85 % inter units, alone and 16 in flight"""
    setup(lib)
    works = works_of(W, H, 32, 9, None, 0.85)
    inputs = b_picture_inputs(W, H)
    nl, units = S.lcu_count(W, H), int(works["num_cus"].sum())
    alone, _ = run_content(lib, root, W, H, works, 1, iters, inputs)
    many, _ = run_content(lib, root, W, H, works, 16, iters, inputs)
    works_t = works_of(W, H, 32, 9, None, 0.85, 0.1)   # what a B picture of an encode looks like: intra units in few LCUs
    alone_t, _ = run_content(lib, root, W, H, works_t, 1, iters, inputs)
    many_t, _ = run_content(lib, root, W, H, works_t, 4, iters, inputs)
    chain = None
    try:
        chain = finish_chain_ms(lib, root, W, H, works, inputs)
    except Exception as e:   # the chain is an extra: never fail the leg for it
        chain = {"error": str(e)[-200:]}
    return {"finished_reference_picture": chain,"content": "%dx%d B picture, random unit trees 8..32, 85 %% inter units (40 %% bi-predicted, random motion within +-24 samples), two "
                       "reference pictures and the work / result arrays resident in HBM: svt_amd_encode_picture_device, ONE launch per picture" % (W, H),
            "lcus": nl, "units": units, "ms_per_picture_alone": round(alone * 1e3, 2), "pictures_per_s_16_in_flight": round(16 / many, 1),
            "lcus_per_s_16_in_flight": round(16 * nl / many),
            "intra_units_in_10_percent_of_the_lcus": {"ms_per_picture_alone": round(alone_t * 1e3, 2), "pictures_per_s_4_in_flight": round(4 / many_t, 1),
                                                      "what": "the same picture with the intra units confined to 10 % of the LCUs: an LCU without intra units waits for no "
                                                              "neighbour (svt_amd_encode_picture_device_inter)"}}


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    lib = S.load_product()
    setup(lib)
    root = C.c_void_p()
    assert lib.svt_amd_context_create(0, W, (H + 7) & ~7, 1, C.byref(root)) == 0, lib.svt_amd_last_error()
    nl = S.lcu_count(W, H)
    rows = []
    inputs = b_picture_inputs(W, H)
    for label, only, inter, il in (("random trees 8..32", None, 0.0, 1.0), ("all 32x32", 32, 0.0, 1.0), ("all 8x8", 8, 0.0, 1.0),
                                   ("B picture: random trees, 85 % inter units", None, 0.85, 1.0),
                                   ("B picture: random trees, inter units; 10 % of the LCUs hold intra units (15 % of theirs)", None, 0.85, 0.1)):
        works = works_of(W, H, 32, 9, only, inter, il)
        units = int(works["num_cus"].sum())
        for P in (1, 4, 16):
            dt, prof = run_content(lib, root, W, H, works, P, iters, inputs if inter else None, profile=P == 1)
            alg = nl * (6144 + 12288 + 6144 + 6144)   # source read, coefficients + LCU reconstruction + picture written
            if prof is not None:
                span = int(prof[:, 6].max() - prof[:, 5].min())
                print("   clocks per LCU (mean): predict %d encode %d copy-out %d wait %d; per unit predict %d encode %d; kernel span %d clocks" %
                      (prof[:, 0].mean(), prof[:, 1].mean(), prof[:, 2].mean(), prof[:, 4].mean(), prof[:, 0].sum() / prof[:, 3].sum(),
                       prof[:, 1].sum() / prof[:, 3].sum(), span), flush=True)
                print("   prediction per unit: availability %d substitution %d smoothing %d samples %d" %
                      tuple(prof[:, 8 + i].sum() / prof[:, 3].sum() for i in range(4)), flush=True)
            rows.append({"content": label, "pictures_in_flight": P, "lcus": nl, "units": units, "ms_per_round": round(dt * 1e3, 3),
                         "pictures_per_s": round(P / dt, 1), "lcus_per_s": round(P * nl / dt), "algorithmic_GBps": round(P * alg / dt / 1e9, 2)})
            print(rows[-1], flush=True)
    print(json.dumps({"width": W, "height": H, "iters": iters, "rows": rows}))


if __name__ == "__main__":
    main()
