"""The closed loop across ranks (SURVEY 8e "EncDec with tiles"): a rank encodes its rectangle of whole tiles, runs the in-loop
filters, ranks exchange their rectangles of the finished picture, and the NEXT pictures predict from the gathered planes - I -> B -> b in coding
order on a recorded random-access encode with two tile columns (tests/golden/encodepass_sao_b_tiles_*.npz).

CPU part (gloo, world size 2): the pinned CPU checker stands in for the device - each process encodes ONLY its rectangle's LCUs, exchanges with
all_gather in the device's slot layout, and every rank ends every picture with the reference encoder's reconstruction, every padded reference with
the reference picture the encoder used.  It proves the split itself: nothing a rank needs from another rank's tiles before the exchange.
GPU part (-m gpu): the same composition through the C-ABI on ONE GPU with two logical ranks (two contexts, two picture objects per picture):
svt_amd_encode_picture_rect -> svt_amd_encdec_picture_deblock -> svt_amd_encdec_picture_sao -> svt_amd_encdec_picture_pack (own rectangle into the
shared slot buffer = what the all-gather delivers, the other's out of it) -> svt_amd_encdec_picture_reference -> svt_amd_encdec_picture_set_inter
of the later pictures.  With RCCL (one process per GPU) svt_amd_encdec_picture_exchange replaces the two pack calls."""
import ctypes as C
import os

import numpy as np
import pytest

import svtlib as S
from test_recon_exchange import Rect, partition, slot_of, unslot

CASE = "sao_b_tiles_motion_512x320_m6"
NONE = 0xFFFFFFFFFFFFFFFF


def tiles_of(g):
    a = g["enc_args"].tolist()
    return int(a[a.index("-tile_col_cnt") + 1]), int(a[a.index("-tile_row_cnt") + 1])


def coding_order(g, nl):
    """pictures in an order in which every picture's reference pictures come first"""
    firsts = {int(g["picture_number"][k]): k for k in range(0, len(g["work"]), nl)}
    done = []
    while len(done) < len(firsts):
        ready = [f for f, k in firsts.items() if f not in done and all(int(v) == NONE or int(v) in done for v in g["ref_poc"][k])]
        assert ready
        done.append(min(ready))
    return [(f, firsts[f]) for f in done]


def lcus_of(rect, w, h):
    wl = (w + 63) // 64
    return [y * wl + x for y in range(rect.y // 64, (rect.y + rect.h + 63) // 64) for x in range(rect.x // 64, (rect.x + rect.w + 63) // 64)]


def finish_picture(oracle, g, w, h, f, first, works, rec, got, mine):
    """deblocking and SAO of picture f behind its encode pass (rec: the planes as encoded, got: the LCUs' results), statistics and decisions of the LCUs in `mine`;
    returns the finished planes"""
    from test_oracle_dlf_golden import oracle_bs, oracle_dlf, oracle_sao
    from test_oracle_encodepass_golden import allows_mismatch, deblock_maps, encoder_order_lcu, is16, sao_inputs_of_picture
    from test_oracle_saodec_golden import STATS, oracle_decide_picture, same_decision
    vp, u32 = C.c_void_p, C.c_uint32
    oracle.svt_oracle_GatherSaoStatistics.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    oracle.svt_oracle_GatherSaoStatistics.restype = None
    bps = 2 if is16(g) else 1
    nl = S.lcu_count(w, h)
    cols, rows = (w + 63) // 64, (h + 63) // 64
    cumap, cbf, qp, edge = deblock_maps(works, got, w, h)
    hdr = dict(width=w, height=h, bytes_per_sample=bps, qp_stride=w // 8, tc_offset=0, beta_offset=0, cb_qp_offset=0, cr_qp_offset=0,
               slice_type=int(works[0]["slice_type"]))
    pic = dict(hdr=hdr, cumap=cumap.reshape(-1), cbf=cbf.reshape(-1), refpoc=np.ascontiguousarray(g["ref_poc"][first]), lcu_edge=edge,
               bsv=np.zeros((nl, 256), np.uint8), bsh=np.zeros((nl, 256), np.uint8))
    pic["bsv"], pic["bsh"] = oracle_bs(oracle, pic)
    pic["pre"], pic["qp"] = rec, qp.reshape(-1)
    mismatch = allows_mismatch(g, w, h, works[0])
    fin = [r.copy() for r in rec] if mismatch else oracle_dlf(oracle, pic)
    P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
    out = fin
    if P is not None:
        stats = np.zeros((3, nl), STATS)
        ncomp = 3 if P["mm_sao"][0] else (1 if P["temporal_layer"][0] < 2 else 0)
        for k in mine:
            wk = works[k]
            x0, y0 = int(wk["lcu_x"]), int(wk["lcu_y"])
            lw, lh = min(64, w - x0), min(64, h - y0)
            for p in range(ncomp):
                sh = 1 if p else 0
                blk = np.ascontiguousarray(encoder_order_lcu(rec, fin, p, x0, y0, lw, lh, w, h, not wk["tile_right"],
                                                             k + cols >= nl or not works[k + cols]["tile_top"]))
                src = np.ascontiguousarray(wk[("src_y", "src_cb", "src_cr")[p]].reshape(64 >> sh, 64 >> sh))
                st = stats[p][k:k + 1]
                oracle.svt_oracle_GatherSaoStatistics(bps, 0 if P["mm_sao"][0] else 1, src.ctypes.data, 64 >> sh, blk.ctypes.data, blk.shape[1],
                                                      lw >> sh, lh >> sh, st["boDiff"].ctypes.data, st["boCount"].ctypes.data,
                                                      st["eoDiff"].ctypes.data, st["eoCount"].ctypes.data)
        dec, _ = oracle_decide_picture(oracle, dict(P=P, stats=stats, enable=enable, params=params, cols=cols, rows=rows))
        for i in idx:
            if i in mine:
                assert same_decision(dec[i], want[i]), (f, int(i), dec[i], want[i])
        dec["edge_flags"] = params["edge_flags"]
        if not mismatch:
            out = oracle_sao(oracle, fin, bps, w, h, dec, 1, 1)
    return out


def oracle_rank_sequence(oracle, g, w, h, rects, rank, gather):
    """one rank's whole sequence with the CPU checker; gather(slot bytes) -> every rank's slot.  Returns the number of pictures checked."""
    from test_oracle_dlf_golden import oracle_bs, oracle_dlf, oracle_sao
    from test_oracle_encodepass_golden import (allows_mismatch, compare_lcu, deblock_maps, encoder_order_lcu, inter_oracle_fn, is16,
                                               sao_inputs_of_picture)
    from test_oracle_saodec_golden import STATS, oracle_decide_picture, same_decision
    wide = is16(g)
    fn = inter_oracle_fn(oracle, wide)
    vp, u32 = C.c_void_p, C.c_uint32
    oracle.svt_oracle_GatherSaoStatistics.argtypes = [C.c_int, C.c_int, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    oracle.svt_oracle_GatherSaoStatistics.restype = None
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    bps = 2 if wide else 1
    nl = S.lcu_count(w, h)
    cols, rows = (w + 63) // 64, (h + 63) // 64
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    mine = lcus_of(rects[rank], w, h)
    pitches = (w, w // 2, w // 2)
    pb = (C.c_uint32 * 3)(*pitches)
    refs, keep, checked = {}, [], 0
    for f, first in coding_order(g, nl):
        works = np.ascontiguousarray(g["work"][first:first + nl])
        rec = [np.zeros((hh, p), sdt) for hh, p in zip((h, h // 2, h // 2), pitches)]
        mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
        rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
        got = np.zeros(nl, rdt)
        r0, r1 = (refs.get(int(v)) for v in g["ref_poc"][first])
        cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)]) if f in g["cost_pictures"].tolist() else None
        for k in mine:       # ONLY this rank's LCUs
            fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None,
               cost.ctypes.data if cost is not None else None, works[k:k + 1].ctypes.data, got[k:k + 1].ctypes.data)
            compare_lcu(works[k], g["result"][first + k], got[k], w, h, (CASE, f, k), rec=False)
        out = finish_picture(oracle, g, w, h, f, first, works, rec, got, mine)
        # before the exchange a rank holds its own rectangle of the finished picture and nothing else of value
        full = [np.zeros_like(p) for p in out]
        unslot(full, rects[rank], slot_of(out, rects[rank], bps), sdt)
        slots = gather(slot_of(full, rects[rank], bps))
        for r, sl in enumerate(slots):
            if r != rank:
                unslot(full, rects[r], sl[:rects[r].w * rects[r].h * bps * 3 // 2], sdt)
        for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
            bad = np.argwhere(full[p] != g[nm][f])
            assert len(bad) == 0, (rank, f, nm, len(bad), bad[:4].tolist())
        padded = [np.ascontiguousarray(np.pad(full[p], (((oy >> 1, oy >> 1), (ox >> 1, ox >> 1)) if p else ((oy, oy), (ox, ox))), mode="edge")) for p in range(3)]
        if f in g["ref_pocs"].tolist():
            i = g["ref_pocs"].tolist().index(f)
            for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):
                assert np.array_equal(padded[p].reshape(-1), g[nm][i]), (rank, f, nm)
        keep.append(padded)
        refs[f] = S.RefPicture(padded[0].ctypes.data, padded[1].ctypes.data, padded[2].ctypes.data, sy, sc, ox, oy, rw, rh)
        checked += 1
    return checked


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_oracle_encodepass_golden import load_case
        g, w, h = load_case(CASE)
        lib = C.CDLL(S.PRODUCT_SO)
        tc, tr = tiles_of(g)
        rc, rects, _ = partition(lib, w, h, tc, tr, world)
        assert rc == 0
        slot_bytes = max(r.w * r.h * 3 // 2 for r in rects) * (2 if g["work"].dtype == S.LCU_WORK16_DTYPE else 1)

        def gather(mine):
            send = np.zeros(slot_bytes, np.uint8)
            send[:mine.size] = mine
            recv = [torch.zeros(slot_bytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(recv, torch.from_numpy(send))
            return [t.numpy() for t in recv]

        n = oracle_rank_sequence(S.load_oracle(), g, w, h, rects, rank, gather)
        q.put((rank, n, ""))
    except Exception as e:   # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, -1, traceback.format_exc()[-1500:] + str(e)[-500:]))
    finally:
        dist.destroy_process_group()


def test_tile_sharded_closed_loop_with_exchange_equals_the_encoder_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] == 5 for r in res), res


def test_one_rank_alone_is_the_unsharded_chain(oracle):
    """world size 1 through the same code = the plain chain (the reference point of the sharded run)"""
    from test_oracle_encodepass_golden import load_case
    g, w, h = load_case(CASE)
    rect = Rect(0, 0, w, h)
    assert oracle_rank_sequence(oracle, g, w, h, [rect], 0, lambda s: [s]) == 5


@pytest.mark.gpu
def test_two_logical_ranks_on_one_gpu_chain_a_sequence_through_the_exchange(product):
    import torch
    from test_gpu_encodepass import DeblockParams, sig_picture
    from test_oracle_encodepass_golden import allows_mismatch, compare_lcu, is16, load_case, sao_inputs_of_picture
    from test_oracle_saodec_golden import LCU, same_decision
    lib = product
    sig_picture(lib)
    g, w, h = load_case(CASE)
    assert not is16(g)
    world = 2
    tc, tr = tiles_of(g)
    rc, rects, _ = partition(lib, w, h, tc, tr, world)
    assert rc == 0
    vp = C.c_void_p
    lib.svt_amd_encode_picture_rect.restype, lib.svt_amd_encode_picture_rect.argtypes = C.c_int, [vp, vp, vp, vp, C.POINTER(Rect)]
    lib.svt_amd_encdec_picture_pack.restype = C.c_int
    lib.svt_amd_encdec_picture_pack.argtypes = [vp, vp, C.POINTER(Rect), C.c_int, C.c_int, vp, C.c_size_t, C.c_int]
    dbk = lib.svt_amd_encdec_picture_deblock
    dbk.restype, dbk.argtypes = C.c_int, [vp, vp, vp, vp, C.POINTER(DeblockParams), vp, vp, vp]
    sao = lib.svt_amd_encdec_picture_sao
    sao.restype, sao.argtypes = C.c_int, [vp] * 9
    lib.svt_amd_encdec_picture_reference.restype = C.c_int
    lib.svt_amd_encdec_picture_reference.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_set_inter.restype, lib.svt_amd_encdec_picture_set_inter.argtypes = C.c_int, [vp] * 5
    nl = S.lcu_count(w, h)
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    ctxs = []
    for r in range(world):
        c = vp()
        assert lib.svt_amd_context_create(0, w, h, 1, C.byref(c)) == 0, lib.svt_amd_last_error()
        ctxs.append(c)
    slot_bytes = (max(r.w * r.h * 3 // 2 for r in rects) + 255) & ~255
    slots = torch.zeros(world * slot_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    pics, refs = [], {}          # refs[f][rank]
    try:
        for f, first in coding_order(g, nl):
            works = np.ascontiguousarray(g["work"][first:first + nl])
            mismatch = allows_mismatch(g, w, h, works[0])
            P, enable, params, want, idx = sao_inputs_of_picture(g, f, works, w, h)
            objs = []
            for r in range(world):     # every rank: its rectangle through encode pass, deblocking and SAO, then its slot
                pic = vp()
                assert lib.svt_amd_encdec_picture_create(ctxs[r], w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
                pics.append((r, pic))
                objs.append(pic)
                r0, r1 = (refs.get(int(v), [None] * world)[r] for v in g["ref_poc"][first])
                if r0 or r1:
                    cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)])
                    assert lib.svt_amd_encdec_picture_set_inter(ctxs[r], pic, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None, cost.ctypes.data) == 0, \
                        lib.svt_amd_last_error()
                got = np.zeros(nl, S.LCU_RESULT_DTYPE)
                assert lib.svt_amd_encode_picture_rect(ctxs[r], pic, works.ctypes.data, got.ctypes.data, C.byref(rects[r])) == 0, lib.svt_amd_last_error()
                mine = lcus_of(rects[r], w, h)
                for k in range(nl):
                    if k in mine:
                        compare_lcu(works[k], g["result"][first + k], got[k], w, h, (CASE, f, r, k), rec=False)
                    else:
                        assert not got[k]["cu"]["cbf"].any() and not got[k]["coeff_y"].any()
                prm = DeblockParams()
                prm.slice_type = int(works[0]["slice_type"])
                prm.ref_poc[0], prm.ref_poc[1] = int(g["ref_poc"][first][0]), int(g["ref_poc"][first][1])
                if not mismatch:
                    assert dbk(ctxs[r], pic, works.ctypes.data, got.ctypes.data, C.byref(prm), None, None, None) == 0, lib.svt_amd_last_error()
                    if P is not None:
                        dec = np.zeros(nl, LCU)
                        assert sao(ctxs[r], pic, works.ctypes.data, P.ctypes.data, enable.ctypes.data, dec.ctypes.data, None, None, None) == 0, lib.svt_amd_last_error()
                        for i in idx:
                            if i in mine:
                                assert same_decision(dec[i], want[i]), (f, r, int(i), dec[i], want[i])
                assert lib.svt_amd_encdec_picture_pack(ctxs[r], pic, rects, world, r, slots.data_ptr(), slot_bytes, 1) == 0, lib.svt_amd_last_error()
                lib.svt_amd_synchronize(ctxs[r])
            refs[f] = []
            for r in range(world):     # ... what the all-gather hands every rank: the others' slots into its own picture, then the padding
                for o in range(world):
                    if o != r:
                        assert lib.svt_amd_encdec_picture_pack(ctxs[r], objs[r], rects, world, o, slots.data_ptr(), slot_bytes, 0) == 0, lib.svt_amd_last_error()
                ref = S.RefPicture()
                padded = [np.zeros(((rh + 2 * oy) >> s_, sy >> s_), np.uint8) for s_ in (0, 1, 1)]
                assert lib.svt_amd_encdec_picture_reference(ctxs[r], objs[r], ox, oy, C.byref(ref), *[a.ctypes.data for a in padded]) == 0, lib.svt_amd_last_error()
                for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                    s_ = 1 if p else 0
                    inner = padded[p][(oy >> s_):(oy >> s_) + (h >> s_), (ox >> s_):(ox >> s_) + (w >> s_)]
                    bad = np.argwhere(inner != g[nm][f])
                    assert len(bad) == 0, (f, r, nm, len(bad), bad[:4].tolist())
                if f in g["ref_pocs"].tolist():
                    i = g["ref_pocs"].tolist().index(f)
                    for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):
                        assert np.array_equal(padded[p].reshape(-1), g[nm][i]), (f, r, nm)
                refs[f].append(ref)
    finally:
        for r, pic in pics:
            lib.svt_amd_encdec_picture_destroy(ctxs[r], pic)
        for c in ctxs:
            lib.svt_amd_context_destroy(c)


@pytest.mark.gpu
def test_encode_picture_rect_rejects_a_rectangle_that_cuts_a_tile(product, gpu_ctx):
    from test_oracle_encodepass_golden import load_case
    lib = product
    g, w, h = load_case(CASE)
    nl = S.lcu_count(w, h)
    works = np.ascontiguousarray(g["work"][:nl])
    got = np.zeros(nl, S.LCU_RESULT_DTYPE)
    vp = C.c_void_p
    lib.svt_amd_encode_picture_rect.restype, lib.svt_amd_encode_picture_rect.argtypes = C.c_int, [vp, vp, vp, vp, C.POINTER(Rect)]
    pic = vp()
    assert lib.svt_amd_encdec_picture_create(gpu_ctx, w, h, 1, C.byref(pic)) == 0
    try:
        for bad in (Rect(0, 0, 128, h), Rect(64, 0, 192, h), Rect(0, 0, w, 64), Rect(32, 0, 224, h), Rect(0, 0, w + 64, h)):
            assert lib.svt_amd_encode_picture_rect(gpu_ctx, pic, works.ctypes.data, got.ctypes.data, C.byref(bad)) != 0
    finally:
        lib.svt_amd_encdec_picture_destroy(gpu_ctx, pic)


@pytest.mark.gpu
def test_mode_decision_and_encode_pass_of_a_ranks_rectangle(product):
    """the picture-level mode-decision call restricted to a rank's tile rectangle (svt_amd_encdec_picture_set_rect): on a recorded random-access encode with
    2 x 2 tiles every one of 2 and of 4 ranks decides and encodes ITS LCUs exactly as the reference did - split, modes, vectors, costs of every tested leaf
    (tests/golden/md_b_tiles_*.npz: the reference's ModeDecisionLcu records) - and leaves the other LCUs' records zeroed"""
    from test_gpu_md import md_encode_inter, sig
    from test_oracle_md_golden import compare_md
    lib = product
    sig(lib)
    vp = C.c_void_p
    lib.svt_amd_encdec_picture_set_rect.restype, lib.svt_amd_encdec_picture_set_rect.argtypes = C.c_int, [vp, vp, C.POINTER(Rect)]
    g = np.load(os.path.join(S.GOLDEN_DIR, "md_b_tiles_motion_640x384_m8.npz"))
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = vp(), vp()
    assert lib.svt_amd_context_create(0, w, h, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    try:
        for world in (2, 4):
            rc, rects, _ = partition(lib, w, h, 2, 2, world)
            assert rc == 0
            for k in range(len(g["picture_number"])):
                seen = np.zeros(S.lcu_count(w, h), bool)
                for r in range(world):
                    assert lib.svt_amd_encdec_picture_set_rect(ctx, pic, C.byref(rects[r])) == 0, lib.svt_amd_last_error()
                    out, works, res = md_encode_inter(lib, ctx, pic, g, k, encode=True)
                    mine = np.zeros(len(out), bool)
                    mine[lcus_of(rects[r], w, h)] = True
                    assert not seen[mine].any()
                    seen |= mine
                    want = g["out"][k].copy()
                    compare_md(out[mine], want[mine], "rank %d of %d, picture %d" % (r, world, int(g["picture_number"][k])))
                    assert not out[~mine]["tested"].any() and not works[~mine]["num_cus"].any() and works[mine]["num_cus"].all()
                assert seen.all()
        # a rectangle that cuts a tile is refused by the call that sees the LCUs' tile flags
        bad = Rect(0, 0, 128, h)
        assert lib.svt_amd_encdec_picture_set_rect(ctx, pic, C.byref(bad)) == 0
        lib.svt_amd_md_encode_picture_inter.restype = C.c_int
        with pytest.raises(AssertionError):
            md_encode_inter(lib, ctx, pic, g, 0, encode=True)
        assert lib.svt_amd_encdec_picture_set_rect(ctx, pic, None) == 0
        out, _, _ = md_encode_inter(lib, ctx, pic, g, 0, encode=True)
        compare_md(out, g["out"][0], "whole picture again")
    finally:
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
        lib.svt_amd_context_destroy(ctx)
