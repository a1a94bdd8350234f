"""Picture-level parallelism across ranks (SURVEY 8e last row): the pictures of a mini-GOP are OWNED by different ranks - a rank encodes the
whole of its picture (every LCU, in-loop filters) - and the only traffic is the finished REFERENCE pictures, broadcast by their owner once; the two non-reference
B pictures of the mini-GOP (temporal layer 2) run on different ranks at the same time, each from the references it received, and send nothing.

Recorded random-access encode tests/golden/encodepass_sao_b_tiles_*.npz: coding order 0 (I) -> 4 (base B) -> 2 (layer-1 B, reference) -> 1, 3 (layer-2 B).
CPU part (gloo, world size 2): the pinned CPU checker stands in for the device; the owner of a picture is its index in coding order modulo the world size, so
0, 2, 3 are rank 0's and 4, 1 rank 1's: picture 4 predicts from a picture of the other rank, 2 from one of each, and 1 / 3 - the concurrent pair - each from a
picture of either rank.  Every owner must end its picture with the reference encoder's reconstruction, every rank every reference with the reference picture the
encoder used.  On the device the broadcast is svt_amd_encdec_picture_broadcast (ncclBroadcast of the picture object's finished planes, before
svt_amd_encdec_picture_reference pads them on every rank)."""
import ctypes as C
import os

import numpy as np

import svtlib as S
from test_tile_ranks import CASE, NONE, coding_order, finish_picture


def owner_of(index_in_coding_order, world):
    return index_in_coding_order % world


def picture_rank_sequence(oracle, g, w, h, world, rank, bcast):
    """one rank's part of the sequence; bcast(planes or None, owner) -> the owner's planes on every rank.  Returns (pictures encoded here, references received)."""
    from test_oracle_encodepass_golden import compare_lcu, inter_oracle_fn, is16
    wide = is16(g)
    fn = inter_oracle_fn(oracle, wide)
    sdt, rdt = (np.uint16, S.LCU_RESULT16_DTYPE) if wide else (np.uint8, S.LCU_RESULT_DTYPE)
    nl = S.lcu_count(w, h)
    sy, sc, ox, oy, rw, rh = (int(v) for v in g["ref_geom"])
    pitches = (w, w // 2, w // 2)
    pb = (C.c_uint32 * 3)(*pitches)
    refs, keep, mine_n, received = {}, [], 0, 0
    for idx, (f, first) in enumerate(coding_order(g, nl)):
        own = owner_of(idx, world) == rank
        is_ref = f in g["ref_pocs"].tolist()
        out = None
        if own:
            works = np.ascontiguousarray(g["work"][first:first + nl])
            rec = [np.zeros((hh, p), sdt) for hh, p in zip((h, h // 2, h // 2), pitches)]
            mp = np.full(((h + 3) // 4, (w + 3) // 4 + 3), 0xFF, np.uint8)
            rp = (C.c_void_p * 3)(*[r.ctypes.data for r in rec])
            got = np.zeros(nl, rdt)
            need = [int(v) for v in g["ref_poc"][first] if int(v) != NONE]
            assert all(v in refs for v in need), (rank, f, need, sorted(refs))   # every reference picture arrived before its first reader starts
            r0, r1 = (refs.get(int(v)) for v in g["ref_poc"][first])
            cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)]) if f in g["cost_pictures"].tolist() else None
            for k in range(nl):
                fn(rp, pb, mp.ctypes.data, mp.shape[1], w, h, C.byref(r0) if r0 else None, C.byref(r1) if r1 else None,
                   cost.ctypes.data if cost is not None else None, works[k:k + 1].ctypes.data, got[k:k + 1].ctypes.data)
                compare_lcu(works[k], g["result"][first + k], got[k], w, h, (CASE, f, k), rec=False)
            out = finish_picture(oracle, g, w, h, f, first, works, rec, got, list(range(nl)))
            for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):   # the owner's picture = the encoder's
                assert np.array_equal(out[p], g[nm][f]), (rank, f, nm)
            mine_n += 1
        if not is_ref:
            continue   # a non-reference picture leaves its rank as bitstream only
        full = bcast(out, owner_of(idx, world), [(hh, p) for hh, p in zip((h, h // 2, h // 2), pitches)], sdt)
        received += 0 if own else 1
        padded = [np.ascontiguousarray(np.pad(full[p], (((oy >> 1, oy >> 1), (ox >> 1, ox >> 1)) if p else ((oy, oy), (ox, ox))), mode="edge")) for p in range(3)]
        i = g["ref_pocs"].tolist().index(f)
        for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):            # every rank's copy = the reference picture the encoder used
            assert np.array_equal(padded[p].reshape(-1), g[nm][i]), (rank, f, nm)
        keep.append(padded)
        refs[f] = S.RefPicture(padded[0].ctypes.data, padded[1].ctypes.data, padded[2].ctypes.data, sy, sc, ox, oy, rw, rh)
    return mine_n, received


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_oracle_encodepass_golden import load_case
        g, w, h = load_case(CASE)

        def bcast(planes, owner, shapes, sdt):
            out = []
            for p, shp in enumerate(shapes):
                t = torch.from_numpy(np.ascontiguousarray(planes[p]).view(np.uint8).reshape(-1).copy()) if rank == owner else torch.zeros(shp[0] * shp[1] * np.dtype(sdt).itemsize,
                                                                                                                                   dtype=torch.uint8)
                dist.broadcast(t, src=owner)
                out.append(t.numpy().view(sdt).reshape(shp))
            return out

        n, r = picture_rank_sequence(S.load_oracle(), g, w, h, world, rank, bcast)
        q.put((rank, n, r, ""))
    except Exception as e:   # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, -1, -1, traceback.format_exc()[-1500:] + str(e)[-500:]))
    finally:
        dist.destroy_process_group()


def test_pictures_of_a_mini_gop_on_different_ranks_with_reference_broadcast_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29890 + os.getpid() % 40
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[0] for r in res] == [0, 1], res
    # rank 0 owns 0, 2, 3 and receives the reference picture 4; rank 1 owns 4, 1 and receives 0 and 2: the layer-2 pair 1 / 3 is split over the ranks
    assert res[0][1:3] == (3, 1) and res[1][1:3] == (2, 2), res


def test_one_rank_owns_every_picture(oracle):
    from test_oracle_encodepass_golden import load_case
    g, w, h = load_case(CASE)
    assert picture_rank_sequence(oracle, g, w, h, 1, 0, lambda planes, owner, shapes, sdt: planes) == (5, 0)


# ---- the mode decision across ranks: the layer-2 pair of a mini-GOP decided on different ranks from broadcast references ---------------------------------------
MD_CASE = "b_motion_416x240_m8"   # pictures 1, 3, 5, 7 of a random-access encode: the layer-2 B pictures of two mini-GOPs, references (0,2) (2,4) (4,6) (6,8)


class _Fixture(dict):
    @property
    def files(self):
        return list(self.keys())


def _md_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_oracle_md_golden import compare_md, oracle_md_picture
        g0 = np.load(os.path.join(S.GOLDEN_DIR, "md_%s.npz" % MD_CASE))
        g = _Fixture({k: g0[k] for k in g0.files})
        npic = len(g["picture_number"])
        # the reference pictures exist on rank 0 only (it stands for their owners); every other rank starts with blank planes and gets each reference picture ONCE
        pocs = g["inter"]["ref_poc"]
        have = {}
        for k in range(npic):
            for l in range(2):
                poc = int(pocs[k][l])
                if poc in have:
                    continue
                planes = []
                for nm in ("y", "cb", "cr"):
                    a = np.ascontiguousarray(g0["ref%d_%s" % (l, nm)][k])
                    t = torch.from_numpy(a.copy() if rank == 0 else np.zeros_like(a))
                    dist.broadcast(t, src=0)
                    planes.append(t.numpy())
                have[poc] = planes
        for l in range(2):
            for p, nm in enumerate(("y", "cb", "cr")):
                g["ref%d_%s" % (l, nm)] = np.stack([have[int(pocs[k][l])][p] for k in range(npic)])
        mine = [k for k in range(npic) if k % world == rank]     # the pair of a mini-GOP (pictures 1 / 3, 5 / 7) is split over the ranks
        oracle = S.load_oracle()
        for k in mine:
            out, _ = oracle_md_picture(oracle, g, k)
            compare_md(out, g0["out"][k], "%s picture %d on rank %d" % (MD_CASE, int(g0["picture_number"][k]), rank))
        q.put((rank, [int(g0["picture_number"][k]) for k in mine], ""))
    except Exception as e:   # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, None, traceback.format_exc()[-1500:] + str(e)[-500:]))
    finally:
        dist.destroy_process_group()


def test_layer2_pair_of_a_mini_gop_decided_on_different_ranks_from_broadcast_references_gloo_world2():
    """ModeDecisionLcu of whole pictures (the CPU checker of the device's decision kernel) on two ranks: the two non-reference B pictures of a mini-GOP on different
    ranks, each from reference pictures it received by broadcast; decisions = the reference's records, leaf for leaf"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29840 + os.getpid() % 40
    ps = [ctx.Process(target=_md_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(60)
    assert res[0][1] == [1, 5] and res[1][1] == [3, 7], res


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_logical_ranks_on_one_gpu_own_whole_pictures_and_hand_over_the_references(product):
    """the same schedule through the C-ABI on ONE GPU with two logical ranks (two contexts): the owner runs svt_amd_encode_picture -> _deblock -> _sao on the whole
    picture, a reference picture goes to the other rank's picture object (svt_amd_encdec_picture_import = what svt_amd_encdec_picture_broadcast delivers with one
    process per GPU), both pad it (svt_amd_encdec_picture_reference) and the later pictures of EITHER rank predict from their own copy"""
    from test_gpu_encodepass import DeblockParams, sig_picture
    from test_oracle_encodepass_golden import allows_mismatch, compare_lcu, is16, load_case, sao_inputs_of_picture
    from test_oracle_saodec_golden import LCU, same_decision
    lib = product
    sig_picture(lib)
    g, w, h = load_case(CASE)
    assert not is16(g)
    world = 2
    vp = C.c_void_p
    lib.svt_amd_encdec_picture_import.restype, lib.svt_amd_encdec_picture_import.argtypes = C.c_int, [vp, vp, vp]
    lib.svt_amd_encdec_picture_broadcast.restype, lib.svt_amd_encdec_picture_broadcast.argtypes = C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]
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
    pics, refs, owned = [], {}, [0, 0]          # refs[f][rank]
    try:
        for idx, (f, first) in enumerate(coding_order(g, nl)):
            owner = owner_of(idx, world)
            works = np.ascontiguousarray(g["work"][first:first + nl])
            mismatch = allows_mismatch(g, w, h, works[0])
            P, enable, params, want, idx_sao = sao_inputs_of_picture(g, f, works, w, h)
            objs = []
            for r in range(world):
                pic = vp()
                assert lib.svt_amd_encdec_picture_create(ctxs[r], w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
                pics.append((r, pic))
                objs.append(pic)
            r0, r1 = (refs.get(int(v), [None] * world)[owner] for v in g["ref_poc"][first])    # the OWNER's copies of the reference pictures
            if r0 or r1:
                cost = np.ascontiguousarray(g["cost"][g["cost_pictures"].tolist().index(f)])
                assert lib.svt_amd_encdec_picture_set_inter(ctxs[owner], objs[owner], C.byref(r0) if r0 else None, C.byref(r1) if r1 else None, cost.ctypes.data) == 0, \
                    lib.svt_amd_last_error()
            got = np.zeros(nl, S.LCU_RESULT_DTYPE)
            assert lib.svt_amd_encode_picture(ctxs[owner], objs[owner], works.ctypes.data, got.ctypes.data) == 0, lib.svt_amd_last_error()
            for k in range(nl):
                compare_lcu(works[k], g["result"][first + k], got[k], w, h, (CASE, f, owner, k), rec=False)
            prm = DeblockParams()
            prm.slice_type = int(works[0]["slice_type"])
            prm.ref_poc[0], prm.ref_poc[1] = int(g["ref_poc"][first][0]), int(g["ref_poc"][first][1])
            if not mismatch:
                assert dbk(ctxs[owner], objs[owner], works.ctypes.data, got.ctypes.data, C.byref(prm), None, None, None) == 0, lib.svt_amd_last_error()
                if P is not None:
                    dec = np.zeros(nl, LCU)
                    assert sao(ctxs[owner], objs[owner], works.ctypes.data, P.ctypes.data, enable.ctypes.data, dec.ctypes.data, None, None, None) == 0, lib.svt_amd_last_error()
                    for i in idx_sao:
                        assert same_decision(dec[i], want[i]), (f, int(i), dec[i], want[i])
            lib.svt_amd_synchronize(ctxs[owner])
            owned[owner] += 1
            holders = range(world) if f in g["ref_pocs"].tolist() else [owner]
            refs[f] = [None] * world
            for r in holders:
                if r != owner:
                    assert lib.svt_amd_encdec_picture_import(ctxs[r], objs[r], objs[owner]) == 0, lib.svt_amd_last_error()
                ref = S.RefPicture()
                padded = [np.zeros(((rh + 2 * oy) >> s_, sy >> s_), np.uint8) for s_ in (0, 1, 1)]
                assert lib.svt_amd_encdec_picture_reference(ctxs[r], objs[r], ox, oy, C.byref(ref), *[a_.ctypes.data for a_ in padded]) == 0, lib.svt_amd_last_error()
                for p, nm in enumerate(("recon_y", "recon_cb", "recon_cr")):
                    s_ = 1 if p else 0
                    inner = padded[p][(oy >> s_):(oy >> s_) + (h >> s_), (ox >> s_):(ox >> s_) + (w >> s_)]
                    assert np.array_equal(inner, g[nm][f]), (f, r, nm)
                if f in g["ref_pocs"].tolist():
                    i = g["ref_pocs"].tolist().index(f)
                    for p, nm in enumerate(("ref_y", "ref_cb", "ref_cr")):
                        assert np.array_equal(padded[p].reshape(-1), g[nm][i]), (f, r, nm)
                refs[f][r] = ref
        assert owned == [3, 2]
        # one rank alone: the collective form is a no-op
        assert lib.svt_amd_encdec_picture_broadcast(ctxs[0], pics[0][1], 1, 0, 0) == 0
    finally:
        for r, pic in pics:
            lib.svt_amd_encdec_picture_destroy(ctxs[r], pic)
        for c in ctxs:
            lib.svt_amd_context_destroy(c)
