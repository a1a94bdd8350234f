"""GPU: the device-resident mode decision + encode pass of whole pictures (svt_amd_md_encode_picture, through the C ABI) against
 (i) the reference's own ModeDecisionLcu records of whole pictures (tests/golden/md_*.npz: split flag, mode, luma cbf and cost of every
     leaf the reference tested),
 (ii) the EncDec input contract those decisions amount to, and
 (iii) the pinned CPU oracle of the encode pass run on that contract in raster order (coefficients, flags, reconstruction)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S
from test_oracle_md_golden import CASES, compare_md, works_from_md, oracle_md_picture
from test_oracle_encodepass_golden import compare_lcu

pytestmark = pytest.mark.gpu


def sig(lib):
    vp = C.c_void_p
    lib.svt_amd_md_encode_picture.restype = C.c_int
    lib.svt_amd_md_encode_picture.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_md_picture_supported.restype = C.c_int
    lib.svt_amd_md_picture_supported.argtypes = [vp]
    lib.svt_amd_encdec_picture_create.restype = C.c_int
    lib.svt_amd_encdec_picture_create.argtypes = [vp, C.c_uint16, C.c_uint16, C.c_int, C.POINTER(vp)]
    lib.svt_amd_encdec_picture_destroy.argtypes = [vp, vp]


def oracle_encode_picture(oracle, works, w, h):
    oracle.svt_oracle_encode_lcu.restype = None
    oracle.svt_oracle_encode_lcu.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p]
    pitches = (w + 32, w // 2 + 16, w // 2 + 16)
    pb = (C.c_uint32 * 3)(*pitches)
    rec = [np.full((hh, p), 0xA5, np.uint8) for hh, p in zip((h, h // 2, h // 2), pitches)]
    mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
    rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
    want = np.zeros(len(works), S.LCU_RESULT_DTYPE)
    for k in range(len(works)):
        oracle.svt_oracle_encode_lcu(rp, pb, mp.ctypes.data, mp.shape[1], w, h, works[k:k + 1].ctypes.data, want[k:k + 1].ctypes.data)
    return want


def md_encode(lib, ctx, pic, g, k, ois=True):
    P = np.ascontiguousarray(g["pic"][k:k + 1])
    lcus = np.ascontiguousarray(g["lcu"][k])
    cost = np.ascontiguousarray(g["cost"][k])
    src = [np.ascontiguousarray(g[n][k]) for n in ("src_y", "src_cb", "src_cr")]
    o = np.ascontiguousarray(g["ois"][k])
    n = len(lcus)
    out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)
    rc = lib.svt_amd_md_encode_picture(ctx, pic, P.ctypes.data, lcus.ctypes.data, src[0].ctypes.data, src[0].shape[1], src[1].ctypes.data,
                                       src[2].ctypes.data, src[1].shape[1], o.ctypes.data if ois else None, 0, cost.ctypes.data, out.ctypes.data,
                                       works.ctypes.data, res.ctypes.data)
    assert rc == 0, lib.svt_amd_last_error()
    return out, works, res


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("i_")])
def test_md_encode_picture_matches_the_reference_and_the_oracle(product, oracle, name):
    lib = product
    sig(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for k in range(len(g["picture_number"])):
            assert lib.svt_amd_md_picture_supported(np.ascontiguousarray(g["pic"][k:k + 1]).ctypes.data) == 1
            for rep in range(2):   # twice: the picture object's maps and completion flags of the first call must not leak into the second
                out, works, res = md_encode(lib, ctx, pic, g, k)
                tag = "%s picture %d call %d" % (name, int(g["picture_number"][k]), rep)
                compare_md(out, g["out"][k], tag)                                  # (i) the reference's decisions
                want_works = works_from_md(g["pic"][k], g["lcu"][k], out, (g["src_y"][k], g["src_cb"][k], g["src_cr"][k]))
                for i in range(len(works)):                                          # (ii) the contract record
                    n = int(want_works[i]["num_cus"])
                    assert int(works[i]["num_cus"]) == n, (tag, i)
                    for f in ("x", "y", "size", "pred_mode", "intra_luma_mode", "bottom_left_ok", "top_right_ok", "qp", "chroma_qp", "leaf_index"):
                        assert np.array_equal(works[i]["cu"][f][:n], want_works[i]["cu"][f][:n]), (tag, i, f)
                    for f in ("src_y", "src_cb", "src_cr", "lcu_x", "lcu_y", "tile_left", "tile_top", "tile_right", "slice_type", "strong_smoothing",
                              "constrained_intra", "full_lambda", "luma_cbf_bits", "pm_core"):
                        assert np.array_equal(works[i][f], want_works[i][f]), (tag, i, f)
                want = oracle_encode_picture(oracle, works, w, h)                    # (iii) the encode pass behind it
                for i in range(len(works)):
                    compare_lcu(works[i], want[i], res[i], w, h, (tag, i))
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


def md_encode_inter(lib, ctx, pic, g, k, encode=False):
    """svt_amd_md_encode_picture_inter on picture k of a P / B fixture: reference pictures into HBM, rate tables, inter inputs"""
    import torch
    from test_oracle_md_golden import inter_inputs
    vp = C.c_void_p
    lib.svt_amd_md_encode_picture_inter.restype = C.c_int
    lib.svt_amd_md_encode_picture_inter.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_md_picture_supported_inter.restype = C.c_int
    lib.svt_amd_md_picture_supported_inter.argtypes = [vp, vp]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [vp] * 5
    P = np.ascontiguousarray(g["pic"][k:k + 1])
    lcus = np.ascontiguousarray(g["lcu"][k])
    cost = np.ascontiguousarray(g["cost"][k])
    src = [np.ascontiguousarray(g[n][k]) for n in ("src_y", "src_cb", "src_cr")]
    o = np.ascontiguousarray(g["ois"][k])
    X, me, tmvp, refs, planes = inter_inputs(g, k)
    assert lib.svt_amd_md_picture_supported_inter(P.ctypes.data, X.ctypes.data) == 1
    dev = [[torch.from_numpy(a).cuda() for a in pl] for pl in planes]
    torch.cuda.synchronize()
    rs = [S.RefPicture(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), r.strideY, r.strideC, r.originX, r.originY, r.width, r.height) for d, r in zip(dev, refs)]
    assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
    n = len(lcus)
    out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)
    rc = lib.svt_amd_md_encode_picture_inter(ctx, pic, P.ctypes.data, X.ctypes.data, lcus.ctypes.data, src[0].ctypes.data, src[0].shape[1], src[1].ctypes.data,
                                             src[2].ctypes.data, src[1].shape[1], o.ctypes.data, 0, me.ctypes.data, 0, tmvp.ctypes.data if tmvp is not None else None,
                                             out.ctypes.data, works.ctypes.data if encode else None, res.ctypes.data if encode else None)
    assert rc == 0, lib.svt_amd_last_error()
    del dev
    return out, works, res


INTER_CASES = [c for c in CASES if c.startswith(("b_", "p_", "bref_", "pref_"))]   # bref_ / pref_: reference pictures, chroma level 4 (CHROMA_MODE_FULL LCUs)


@pytest.mark.parametrize("name", INTER_CASES)
def test_md_of_p_and_b_pictures_matches_the_reference(product, name):
    """P / B pictures: motion-estimation candidates, AMVP and merge lists (spatial neighbours from the picture's motion-vector map, the temporal
    candidate from the co-located picture's motion field), motion-compensated and open-loop intra candidates, fast and full loops with partial
    frequency N2, merge / skip costs, stop-split and small-unit skips, inter-depth decisions - against the reference's ModeDecisionLcu records"""
    lib = product
    sig(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for k in range(len(g["picture_number"])):
            for rep in range(2):
                out, _, _ = md_encode_inter(lib, ctx, pic, g, k)
                compare_md(out, g["out"][k], "%s picture %d call %d" % (name, int(g["picture_number"][k]), rep))
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("name", [c for c in INTER_CASES if c.startswith(("b_motion", "bref_motion", "p_"))][:4])
def test_md_of_p_and_b_pictures_with_the_transforms_on_the_register_butterflies(product, name):
    """the full loops' 16x16 / 32x32 forward transforms run on the matrix cores (exact f16 x f16 -> f32 integer products) and fall back to the register butterflies for a
    unit outside the Estimate butterflies' wrap-free domain - which no fixture reaches by itself.  svt_amd_debug_md_force_butterflies sends EVERY unit down that path: the
    decisions must be the reference's both ways (so the two transform forms agree with each other on every unit of the fixtures)"""
    lib = product
    sig(lib)
    lib.svt_amd_debug_md_force_butterflies.restype = C.c_int
    lib.svt_amd_debug_md_force_butterflies.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for on in (1, 0):
            assert lib.svt_amd_debug_md_force_butterflies(ctx, pic, on) == 0, lib.svt_amd_last_error()
            for k in range(len(g["picture_number"])):
                out, _, _ = md_encode_inter(lib, ctx, pic, g, k)
                compare_md(out, g["out"][k], "%s picture %d, butterflies forced: %d" % (name, int(g["picture_number"][k]), on))
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("name", INTER_CASES)
def test_md_and_encode_pass_of_p_and_b_pictures_in_one_call(product, oracle, name):
    """the same call with the encode pass behind the decisions: the work records (final tree, vectors, and for merge units the merge / skip decision
    EncodePass makes from the chroma-completed costs - AddChromaEncDec on the device) against the reference's EncodePass records of the same encode,
    and the encode pass's output against the pinned CPU oracle run on those records in raster order"""
    from test_oracle_encodepass_golden import inter_oracle_fn
    from test_oracle_md_golden import inter_inputs, compare_kinds
    lib = product
    sig(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    fn = inter_oracle_fn(oracle, False)
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for k in range(len(g["picture_number"])):
            tag = "%s picture %d" % (name, int(g["picture_number"][k]))
            out, works, res = md_encode_inter(lib, ctx, pic, g, k, encode=True)
            compare_md(out, g["out"][k], tag)
            kinds = np.full((len(works), 85), 0xFF, np.uint8)
            for i in range(len(works)):
                n = int(works[i]["num_cus"])
                cu = works[i]["cu"][:n]
                inter = cu["pred_mode"] == 1
                kinds[i, cu["leaf_index"][inter]] = cu["inter_kind"][inter]
                assert np.array_equal(cu["mv"][inter], g["out"][k][i]["mv"][cu["leaf_index"][inter]]), (tag, i)
                assert np.array_equal(cu["inter_dir"][inter], g["out"][k][i]["inter_dir"][cu["leaf_index"][inter]]), (tag, i)
            compare_kinds(kinds, g["ep_kind"][k], tag)
            # the encode pass behind it: the oracle on the device's own work records, raster order
            X, me, tmvp, refs, planes = inter_inputs(g, k)
            cost = np.ascontiguousarray(g["cost"][k])
            pitches = (w + 32, w // 2 + 16, w // 2 + 16)
            pb = (C.c_uint32 * 3)(*pitches)
            rec = [np.full((hh, p), 0xA5, np.uint8) for hh, p in zip((h, h // 2, h // 2), pitches)]
            mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
            rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
            want = np.zeros(len(works), S.LCU_RESULT_DTYPE)
            for i in range(len(works)):
                fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(refs[0]), C.byref(refs[1]), cost.ctypes.data, works[i:i + 1].ctypes.data,
                   want[i:i + 1].ctypes.data)
                compare_lcu(works[i], want[i], res[i], w, h, (tag, i))
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


def test_md_encode_picture_reads_the_ois_records_the_front_half_left_in_hbm(product, oracle):
    """ois == NULL: the open-loop intra search of the picture ran on the device (svt_amd_ois_picture) and its records are read where they are"""
    lib = product
    sig(lib)
    name = "i_motion_1920x1080_m10" if "i_motion_1920x1080_m10" in CASES else CASES[0]   # encMode 10: the candidate lists come from the OIS records
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    from gpu_util import upload
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        upload(lib, ctx, 0, np.ascontiguousarray(g["src_y"][0]))
        op = S.OisParams()
        op.luma_width, op.luma_height, op.slice_is_intra, op.temporal_layer_index = w, h, 1, 0
        op.skip_ois_8x8, op.cu8x8_mode = int(g["pic"][0]["skip_ois_8x8"]), int(g["pic"][0]["cu8x8_mode"])
        op.limit_ois_to_dc_mode, op.ois_th_set = int(g["pic"][0]["limit_ois_to_dc_mode"]), 1
        ois = np.zeros(S.lcu_count(w, h), S.OIS_LCU_DTYPE)
        assert lib.svt_amd_ois_picture(ctx, C.byref(op), 0, None, ois.ctypes.data) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        out, works, res = md_encode(lib, ctx, pic, g, 0, ois=False)
        compare_md(out, g["out"][0], name + " (device OIS records)")
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


def test_md_of_a_b_picture_reads_me_and_ois_records_another_lane_left_in_hbm(product):
    """me == NULL and ois == NULL: motion estimation and open-loop intra search of the picture were LAUNCHED (not fetched) on another lane's stream; the mode decision
    reads their records where they are, ordered behind those kernels by the slot's events.  Same decisions as with the records passed as host arrays;
    a slot that holds no such records is refused."""
    import torch
    from gpu_util import upload, default_params
    from test_oracle_md_golden import inter_inputs
    lib = product
    sig(lib)
    vp = C.c_void_p
    lib.svt_amd_md_encode_picture_inter.restype = C.c_int
    lib.svt_amd_md_encode_picture_inter.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [vp] * 5
    lib.svt_amd_context_fork.argtypes = [vp, C.POINTER(vp)]
    lib.svt_amd_me_picture_range_launch.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, C.c_uint32]
    lib.svt_amd_ois_picture_launch.argtypes = [vp, vp, C.c_int]
    lib.svt_amd_me_picture_fetch.argtypes = [vp, C.c_int, vp]
    lib.svt_amd_ois_picture_fetch.argtypes = [vp, C.c_int, vp]
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_bref_motion_416x240_m8.npz"))   # a reference B picture: CHROMA_MODE_FULL LCUs
    k = 0
    w, h = int(g["pic"][k]["width"]), int(g["pic"][k]["height"])
    nl = S.lcu_count(w, h)
    root, lane_a, lane_b, pic = vp(), vp(), vp(), vp()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 4, C.byref(root)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_context_fork(root, C.byref(lane_a)) == 0 and lib.svt_amd_context_fork(root, C.byref(lane_b)) == 0, lib.svt_amd_last_error()
        frames = [S.gen_luma("motion", w, h, t, 7) for t in range(3)]
        for sl, f in enumerate(frames):
            upload(lib, lane_a, sl, f)
        mp = default_params(w, h, num_lists=2, temporal_layer_index=1, cu8x8_mode=0)
        refs = (C.c_int * 2)(0, 2)
        op = S.OisParams()
        op.luma_width, op.luma_height, op.ois_th_set, op.temporal_layer_index = w, h, 1, 1
        # lane A: launches only - its kernels may still run when lane B's call arrives
        assert lib.svt_amd_me_picture_range_launch(lane_a, C.byref(mp), 1, refs, 0, nl) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_ois_picture_launch(lane_a, C.byref(op), 1) == 0, lib.svt_amd_last_error()
        P, lcus, cost = np.ascontiguousarray(g["pic"][k:k + 1]), np.ascontiguousarray(g["lcu"][k]), np.ascontiguousarray(g["cost"][k])
        src = [np.ascontiguousarray(g[n][k]) for n in ("src_y", "src_cb", "src_cr")]
        X, _, tmvp, rr, planes = inter_inputs(g, k)
        dev = [[torch.from_numpy(a).cuda() for a in pl] for pl in planes]
        torch.cuda.synchronize()
        rs = [S.RefPicture(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), r.strideY, r.strideC, r.originX, r.originY, r.width, r.height) for d, r in zip(dev, rr)]
        assert lib.svt_amd_encdec_picture_create(lane_b, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_encdec_picture_set_inter(lane_b, pic, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()

        def call(me, ois, slot):
            out = np.zeros(nl, S.MD_LCU_OUT_DTYPE)
            rc = lib.svt_amd_md_encode_picture_inter(lane_b, pic, P.ctypes.data, X.ctypes.data, lcus.ctypes.data, src[0].ctypes.data, src[0].shape[1], src[1].ctypes.data,
                                                     src[2].ctypes.data, src[1].shape[1], ois, slot, me, slot, tmvp.ctypes.data if tmvp is not None else None, out.ctypes.data,
                                                     None, None)
            return rc, out
        rc, in_place = call(None, None, 1)                  # lane B: records of slot 1 read where lane A's kernels leave them
        assert rc == 0, lib.svt_amd_last_error()
        me_h, ois_h = np.zeros(nl, S.ME_LCU_DTYPE), np.zeros(nl, S.OIS_LCU_DTYPE)
        assert lib.svt_amd_me_picture_fetch(lane_a, 1, me_h.ctypes.data) == 0 and lib.svt_amd_ois_picture_fetch(lane_a, 1, ois_h.ctypes.data) == 0, lib.svt_amd_last_error()
        rc, from_host = call(me_h.ctypes.data, ois_h.ctypes.data, 0)
        assert rc == 0, lib.svt_amd_last_error()
        compare_md(in_place, from_host, "records read in place vs the same records as host arrays")   # (every tested leaf; untested leaves hold no decision)
        assert int(in_place["tested"].sum()) >= nl
        rc, _ = call(None, None, 3)                         # a slot no motion estimation / intra search has written
        assert rc != 0
        del dev
        lib.svt_amd_encdec_picture_destroy(lane_b, pic)
    finally:
        for c in (lane_a, lane_b, root):
            if c:
                lib.svt_amd_context_destroy(c)


def test_md_encode_picture_rejects_what_it_does_not_cover(product):
    lib = product
    sig(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % CASES[0]))
    P = np.ascontiguousarray(g["pic"][0:1]).copy()
    for field, value in (("slice_type", 1), ("chroma_level", 0), ("coeff_cabac_update", 1), ("intra_md_open_loop", 1), ("depth_mode", 0)):
        Q = P.copy()
        Q[0][field] = value
        assert lib.svt_amd_md_picture_supported(Q.ctypes.data) == 0, field


@pytest.mark.parametrize("w,h,enc_mode", [(3840, 2160, 7), (1920, 1080, 9)])
def test_md_encode_picture_at_baseline_sizes_matches_the_reference_run_on_the_box(product, w, h, enc_mode):
    """BASELINE configs[2]'s I picture (4K, encMode 7) and configs[1]'s (1080p, encMode 9): the prebuilt reference (oracle/_ref) encodes the picture
    on this box with the recording harness on; the device's decisions for the recorded inputs must be the reference's, leaf for leaf"""
    import sys
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import md_bench
    if not os.path.exists(S.REF_APP):
        pytest.skip("oracle/_ref not built")
    g = md_bench.record(w, h, enc_mode)
    r = md_bench.run(product, g, reps=1)
    assert r["lcus"] == S.lcu_count(w, h) and r["final_units"] > r["lcus"]


@pytest.mark.parametrize("w,h,enc_mode,frames", [(3840, 2160, 7, 5), (1920, 1080, 8, 5)])
def test_md_of_b_pictures_at_baseline_sizes_matches_the_reference_run_on_the_box(product, w, h, enc_mode, frames):
    """BASELINE configs[2] (4K encMode 7 random access) and a 1080p twin: the prebuilt reference encodes `frames` pictures on this box with the
    recording harness on; mode decision + merge / skip decisions + encode pass of its non-reference B pictures in ONE device call each, the
    decisions (every tested leaf: split, mode, vectors, merge index, costs) the reference's"""
    import sys
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import md_bench
    if not os.path.exists(S.REF_APP):
        pytest.skip("oracle/_ref not built")
    g = md_bench.record_inter(w, h, enc_mode, frames=frames, levels=2, ref=None)   # temporal layer 2 (luma-only candidates) and layer 1 (CHROMA_MODE_FULL)
    assert len(g["picture_number"]) >= 3 and g["pic"]["is_reference"].any() and not g["pic"]["is_reference"].all()
    r = md_bench.run_inter(product, g, reps=1, encode=True)
    assert r["lcus"] == S.lcu_count(w, h) and r["final_units"] >= r["lcus"]


# ---- 10-bit pictures (BASELINE configs[3]) -------------------------------------------------------------------------------------------------
# The reference's mode decision of a 10-bit encode is an 8-bit process: its source is the input picture's 8-bit plane and its inter candidates are
# predicted from the 8 most significant bits of the 16-bit reference pictures (Inter2Nx2NPuPredictionHevc with is16bit: UnPackReferenceBlock,
# Codec/EbInterPrediction.c:414-457, sample >> 2); only EncodePass codes the 10-bit samples.  A 10-bit picture whose samples are the fixture's 8-bit
# ones with two arbitrary extra bits therefore has to be DECIDED exactly as the reference decided the fixture's picture - its records are the
# expectation - and ENCODED as the pinned 16-bit oracle encodes the device's work records.
def _widen(a, rng):
    return ((a.astype(np.uint16) << 2) | rng.integers(0, 4, a.shape, dtype=np.uint16)).astype(np.uint16)


def _sig16(lib):
    vp = C.c_void_p
    lib.svt_amd_md_encode_picture16.restype = C.c_int
    lib.svt_amd_md_encode_picture16.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_md_encode_picture_inter16.restype = C.c_int
    lib.svt_amd_md_encode_picture_inter16.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [vp] * 5


def _check_source16(works, src16, w, h, tag):
    for i in range(len(works)):
        x0, y0 = int(works[i]["lcu_x"]), int(works[i]["lcu_y"])
        lw, lh = min(64, w - x0), min(64, h - y0)
        assert np.array_equal(works[i]["src_y"].reshape(64, 64)[:lh, :lw], src16[0][y0:y0 + lh, x0:x0 + lw]), (tag, i, "src_y")
        for p, nm in ((1, "src_cb"), (2, "src_cr")):
            assert np.array_equal(works[i][nm].reshape(32, 32)[:lh // 2, :lw // 2], src16[p][y0 // 2:(y0 + lh) // 2, x0 // 2:(x0 + lw) // 2]), (tag, i, nm)


@pytest.mark.parametrize("name", ["i_motion_416x240_m9", "i_tiles_motion_640x384_m9"])
def test_md_encode_picture16_decides_on_the_8_msbs_and_encodes_the_10_bit_samples(product, oracle, name):
    lib = product
    sig(lib)
    _sig16(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    rng = np.random.default_rng(10)
    oracle.svt_oracle_encode_lcu16.restype = None
    oracle.svt_oracle_encode_lcu16.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 2, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for k in range(len(g["picture_number"])):
            tag = "%s picture %d (10-bit)" % (name, int(g["picture_number"][k]))
            P, lcus, cost, o = (np.ascontiguousarray(v) for v in (g["pic"][k:k + 1], g["lcu"][k], g["cost"][k], g["ois"][k]))
            src16 = [_widen(g[n][k], rng) for n in ("src_y", "src_cb", "src_cr")]
            n = len(lcus)
            out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK16_DTYPE), np.zeros(n, S.LCU_RESULT16_DTYPE)
            rc = lib.svt_amd_md_encode_picture16(ctx, pic, P.ctypes.data, lcus.ctypes.data, src16[0].ctypes.data, src16[0].shape[1], src16[1].ctypes.data,
                                                 src16[2].ctypes.data, src16[1].shape[1], o.ctypes.data, 0, cost.ctypes.data, out.ctypes.data, works.ctypes.data,
                                                 res.ctypes.data)
            assert rc == 0, lib.svt_amd_last_error()
            compare_md(out, g["out"][k], tag)                                   # the reference's decisions
            _check_source16(works, src16, w, h, tag)
            pitches = (w + 32, w // 2 + 16, w // 2 + 16)                        # the encode pass behind them: the pinned 16-bit oracle, raster order
            pb = (C.c_uint32 * 3)(*pitches)
            rec = [np.full((hh, p), 0xA5A5, np.uint16) for hh, p in zip((h, h // 2, h // 2), pitches)]
            mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
            rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
            want = np.zeros(n, S.LCU_RESULT16_DTYPE)
            for i in range(n):
                oracle.svt_oracle_encode_lcu16(rp, pb, mp.ctypes.data, mp.shape[1], w, h, works[i:i + 1].ctypes.data, want[i:i + 1].ctypes.data)
                compare_lcu(works[i], want[i], res[i], w, h, (tag, i))
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("name", [c for c in INTER_CASES if c in ("b_motion_416x240_m8", "bref_motion_416x240_m8", "pref_motion_416x240_m8_ld", "bref_tiles_motion_640x384_m8")])
def test_md_encode_picture_inter16_decides_on_the_8_msbs_and_encodes_the_10_bit_samples(product, oracle, name):
    import torch
    from test_oracle_encodepass_golden import inter_oracle_fn
    from test_oracle_md_golden import inter_inputs, compare_kinds
    lib = product
    sig(lib)
    _sig16(lib)
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % name))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    rng = np.random.default_rng(11)
    fn = inter_oracle_fn(oracle, True)
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 2, C.byref(pic)) == 0, lib.svt_amd_last_error()
        for k in range(len(g["picture_number"])):
            tag = "%s picture %d (10-bit)" % (name, int(g["picture_number"][k]))
            P, lcus, cost, o = (np.ascontiguousarray(v) for v in (g["pic"][k:k + 1], g["lcu"][k], g["cost"][k], g["ois"][k]))
            X, me, tmvp, refs8, planes8 = inter_inputs(g, k)
            src16 = [_widen(g[n][k], rng) for n in ("src_y", "src_cb", "src_cr")]
            planes16 = [[_widen(a, rng) for a in pl] for pl in planes8]
            refs16 = [S.RefPicture(pl[0].ctypes.data, pl[1].ctypes.data, pl[2].ctypes.data, r.strideY, r.strideC, r.originX, r.originY, r.width, r.height)
                      for pl, r in zip(planes16, refs8)]
            dev = [[torch.from_numpy(a.view(np.int16)).cuda() for a in pl] for pl in planes16]
            torch.cuda.synchronize()
            rs = [S.RefPicture(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), r.strideY, r.strideC, r.originX, r.originY, r.width, r.height) for d, r in zip(dev, refs8)]
            assert lib.svt_amd_encdec_picture_set_inter(ctx, pic, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
            n = len(lcus)
            out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK16_DTYPE), np.zeros(n, S.LCU_RESULT16_DTYPE)
            rc = lib.svt_amd_md_encode_picture_inter16(ctx, pic, P.ctypes.data, X.ctypes.data, lcus.ctypes.data, src16[0].ctypes.data, src16[0].shape[1],
                                                       src16[1].ctypes.data, src16[2].ctypes.data, src16[1].shape[1], o.ctypes.data, 0, me.ctypes.data, 0,
                                                       tmvp.ctypes.data if tmvp is not None else None, out.ctypes.data, works.ctypes.data, res.ctypes.data)
            assert rc == 0, lib.svt_amd_last_error()
            compare_md(out, g["out"][k], tag)                                   # the reference's decisions
            kinds = np.full((n, 85), 0xFF, np.uint8)
            for i in range(n):
                m = int(works[i]["num_cus"])
                cu = works[i]["cu"][:m]
                inter = cu["pred_mode"] == 1
                kinds[i, cu["leaf_index"][inter]] = cu["inter_kind"][inter]
                assert np.array_equal(cu["mv"][inter], g["out"][k][i]["mv"][cu["leaf_index"][inter]]), (tag, i)
            compare_kinds(kinds, g["ep_kind"][k], tag)                          # ... and its merge / skip decisions (AddChromaEncDec is an 8-bit process too)
            _check_source16(works, src16, w, h, tag)
            pitches = (w + 32, w // 2 + 16, w // 2 + 16)
            pb = (C.c_uint32 * 3)(*pitches)
            rec = [np.full((hh, p), 0xA5A5, np.uint16) for hh, p in zip((h, h // 2, h // 2), pitches)]
            mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
            rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
            want = np.zeros(n, S.LCU_RESULT16_DTYPE)
            for i in range(n):
                fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(refs16[0]), C.byref(refs16[1]), cost.ctypes.data, works[i:i + 1].ctypes.data,
                   want[i:i + 1].ctypes.data)
                compare_lcu(works[i], want[i], res[i], w, h, (tag, i))
            del dev
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    finally:
        lib.svt_amd_context_destroy(ctx)


def test_md_after_the_warm_up_with_page_locked_padded_source_planes(product):
    """What the encoder binding does since round 5: the picture object's mode-decision state is made BEFORE the first call (svt_amd_md_picture_warmup), the source planes
    live in page-locked buffers with the encoder's row pitch (svt_amd_host_register: the call fetches them by a kernel of its own stream).  (svt_amd_host_wait_mode is a
    setting of the DEVICE for the whole process: exercised by the encoder bindings in the end-to-end cases, not inside this test process.)  Decisions, work and result records must be those of the plain call on pageable, tightly pitched planes."""
    lib = product
    sig(lib)
    vp = C.c_void_p
    lib.svt_amd_md_picture_warmup.restype, lib.svt_amd_md_picture_warmup.argtypes = C.c_int, [vp, vp]
    lib.svt_amd_host_register.restype, lib.svt_amd_host_register.argtypes = C.c_int, [vp, vp, C.c_size_t]
    lib.svt_amd_host_unregister_all.restype, lib.svt_amd_host_unregister_all.argtypes = C.c_int, [vp]
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "md_b_objects_416x240_m8.npz")))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic, pic2 = vp(), vp(), vp()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    try:
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0 and lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic2)) == 0
        want = md_encode_inter(lib, ctx, pic, g, 0, encode=True)
        assert lib.svt_amd_md_picture_warmup(ctx, pic2) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_md_picture_warmup(ctx, pic2) == 0                       # idempotent
        # padded planes: 48 columns of padding left of the picture, a row pitch of w + 96 (w / 2 + 48 for chroma), the pictures' samples inside
        pads = []
        for n, pw, ph in (("src_y", w, h), ("src_cb", w // 2, h // 2), ("src_cr", w // 2, h // 2)):
            buf = np.full((ph + 8, pw + 96 // (1 if n == "src_y" else 2)), 0x77, np.uint8)
            off = 48 // (1 if n == "src_y" else 2)
            buf[4:4 + ph, off:off + pw] = g[n][0]
            assert lib.svt_amd_host_register(ctx, buf.ctypes.data, buf.nbytes) == 0
            pads.append((buf, buf[4:, off:]))
        P, lcus = np.ascontiguousarray(g["pic"][0:1]), np.ascontiguousarray(g["lcu"][0])
        o = np.ascontiguousarray(g["ois"][0])
        from test_oracle_md_golden import inter_inputs
        import torch
        X, me, tmvp, refs, planes = inter_inputs(g, 0)
        dev = [[torch.from_numpy(a).cuda() for a in pl] for pl in planes]
        torch.cuda.synchronize()
        rs = [S.RefPicture(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), r.strideY, r.strideC, r.originX, r.originY, r.width, r.height) for d, r in zip(dev, refs)]
        cost = np.ascontiguousarray(g["cost"][0])
        assert lib.svt_amd_encdec_picture_set_inter(ctx, pic2, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
        n = len(lcus)
        out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)
        for rep in range(2):
            rc = lib.svt_amd_md_encode_picture_inter(ctx, pic2, P.ctypes.data, X.ctypes.data, lcus.ctypes.data, pads[0][1].ctypes.data, pads[0][0].shape[1],
                                                     pads[1][1].ctypes.data, pads[2][1].ctypes.data, pads[1][0].shape[1], o.ctypes.data, 0, me.ctypes.data, 0,
                                                     tmvp.ctypes.data if tmvp is not None else None, out.ctypes.data, works.ctypes.data, res.ctypes.data)
            assert rc == 0, lib.svt_amd_last_error()
            compare_md(out, g["out"][0], "padded page-locked planes, call %d" % rep)       # the reference's own records (fields of leaves no call tested are unspecified)
            assert np.array_equal(out["split"], want[0]["split"]) and np.array_equal(out["tested"], want[0]["tested"])
            assert works.tobytes() == want[1].tobytes() and res.tobytes() == want[2].tobytes(), rep
        assert lib.svt_amd_host_unregister_all(ctx) == 0
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_encdec_picture_destroy(ctx, pic2)
    finally:
        lib.svt_amd_context_destroy(ctx)
