"""-m gpu: the asynchronous front-end pipeline (lanes = streams over shared picture slots, pinned staging, ME + OIS + result
copies queued without a host wait) gives exactly the records of the blocking per-picture calls and of the oracle; and the
LCU-row-band form of the ME launch (the multi-GPU shard) stitches to the full picture."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from gpu_util import default_params, me_picture, upload

pytestmark = pytest.mark.gpu


def _records(ptr, n, dtype):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * dtype.itemsize,)).view(dtype).copy()


def test_lanes_pipeline_matches_blocking_calls_and_oracle(product, oracle):
    lib = product
    w, h, npic = 448, 328, 6
    nl = S.lcu_count(w, h)
    root = C.c_void_p()
    assert lib.svt_amd_context_create(0, 640, 384, npic, C.byref(root)) == 0, lib.svt_amd_last_error()
    lanes = []
    try:
        for _ in range(3):
            lane = C.c_void_p()
            assert lib.svt_amd_context_fork(root, C.byref(lane)) == 0, lib.svt_amd_last_error()
            lanes.append(lane)
        # start-up warm-up (pinned buffers, code objects, one dummy picture through slot 0) must leave no trace
        assert lib.svt_amd_frontend_warmup(root) == 0, lib.svt_amd_last_error()
        for lane in lanes:
            assert lib.svt_amd_frontend_warmup(lane) == 0, lib.svt_amd_last_error()
        frames = [S.gen_luma("motion", w, h, t, 7) for t in range(npic)]
        # pictures uploaded through DIFFERENT lanes than the ones that search them: cross-stream ordering via the slot events
        for s, f in enumerate(frames):
            f = np.ascontiguousarray(f)
            assert lib.svt_amd_picture_upload_async(lanes[(s + 1) % 3], s, f.ctypes.data, w, w, h) == 0, lib.svt_amd_last_error()
        p = default_params(w, h, num_lists=2, temporal_layer_index=1, cu8x8_mode=0)
        op = S.OisParams()
        op.luma_width, op.luma_height, op.ois_th_set, op.temporal_layer_index = w, h, 1, 1
        jobs = []
        for k, cur in enumerate((1, 2, 3)):  # three B pictures in flight at once, one per lane
            j = S.FrontendJob()
            j.cur_slot, j.has_me, j.has_ois, j.me, j.ois = cur, 1, 1, p, op
            j.ref_slot[0], j.ref_slot[1] = cur - 1, cur + 1
            assert lib.svt_amd_frontend_submit(lanes[k], C.byref(j)) == 0, lib.svt_amd_last_error()
            jobs.append(j)
        # a second submit on a busy lane is refused, not queued over the pinned buffers
        assert lib.svt_amd_frontend_submit(lanes[0], C.byref(jobs[0])) != 0
        pics = [S.OraclePicture(oracle, f) for f in frames]
        for k, cur in enumerate((1, 2, 3)):
            me_p, ois_p = C.c_void_p(), C.c_void_p()
            assert lib.svt_amd_frontend_wait(lanes[k], C.byref(me_p), C.byref(ois_p)) == 0, lib.svt_amd_last_error()
            me = _records(me_p, nl, S.ME_LCU_DTYPE)
            ois = _records(ois_p, nl, S.OIS_LCU_DTYPE)
            assert lib.svt_amd_frontend_release(lanes[k]) == 0
            want = S.oracle_me_picture(oracle, p, pics[cur], pics[cur - 1], pics[cur + 1])
            S.compare_me(me, want, 2, "lane %d" % k)
            want_ois = S.oracle_ois_picture(oracle, op, frames[cur], want)
            assert np.array_equal(ois["candidate"], want_ois["candidate"]) and np.array_equal(ois["total"], want_ois["total"])
            # and the blocking form on the owning context gives the same bytes
            again = me_picture(lib, root, p, cur, [cur - 1, cur + 1])
            assert np.array_equal(again["pu"], me["pu"])
        # an intra picture: OIS only
        oi = S.OisParams()
        oi.luma_width, oi.luma_height, oi.ois_th_set, oi.slice_is_intra = w, h, 1, 1
        j = S.FrontendJob()
        j.cur_slot, j.has_me, j.has_ois, j.ois = 0, 0, 1, oi
        assert lib.svt_amd_frontend_submit(lanes[1], C.byref(j)) == 0, lib.svt_amd_last_error()
        ois_p = C.c_void_p()
        assert lib.svt_amd_frontend_wait(lanes[1], None, C.byref(ois_p)) == 0
        ois = _records(ois_p, nl, S.OIS_LCU_DTYPE)
        want_ois = S.oracle_ois_picture(oracle, oi, frames[0], None)
        assert np.array_equal(ois["candidate"], want_ois["candidate"]) and np.array_equal(ois["total"], want_ois["total"])
        lib.svt_amd_frontend_release(lanes[1])
        # COMPACT wire format: the same two jobs again with job.compact = 1 - MeCuResult[85] per LCU and nc candidates per CU
        # (nc = 9 on a B picture, 7 on an I picture) must be exactly the corresponding parts of the full records
        for k, (job, slot, full_ois) in enumerate(((jobs[1], 2, None), (j, 0, ois))):
            job.compact = 1
            nc = lib.svt_amd_ois_compact_candidates(C.byref(job.ois))
            assert nc == (7 if job.ois.slice_is_intra else 9)
            assert lib.svt_amd_frontend_submit(lanes[k], C.byref(job)) == 0, lib.svt_amd_last_error()
            me_p, ois_p = C.c_void_p(), C.c_void_p()
            assert lib.svt_amd_frontend_wait(lanes[k], C.byref(me_p), C.byref(ois_p)) == 0, lib.svt_amd_last_error()
            cdt = np.dtype([("candidate", "<u4", (S.ME_PU_COUNT, nc)), ("total", "u1", (S.ME_PU_COUNT,)), ("pad", "u1", (3,))])
            assert cdt.itemsize == S.ME_PU_COUNT * nc * 4 + 88
            cois = _records(ois_p, nl, cdt)
            if job.has_me:
                cme = _records(me_p, nl * S.ME_PU_COUNT, S.ME_LCU_DTYPE["pu"].base)
                full = me_picture(lib, root, p, slot, [slot - 1, slot + 1])
                assert cme.tobytes() == np.ascontiguousarray(full["pu"]).tobytes()
                full_ois = S.oracle_ois_picture(oracle, op, frames[slot], S.oracle_me_picture(oracle, p, pics[slot], pics[slot - 1], pics[slot + 1]))
            assert np.array_equal(cois["candidate"], full_ois["candidate"][:, :, :nc]) and np.array_equal(cois["total"], full_ois["total"])
            assert not (full_ois["candidate"][:, :, nc:] & (S.OIS_W_DIST | S.OIS_W_VALID | S.OIS_W_MODE)).any()   # nothing written beyond nc
            lib.svt_amd_frontend_release(lanes[k])
    finally:
        for lane in lanes:
            lib.svt_amd_context_destroy(lane)
        lib.svt_amd_context_destroy(root)


@pytest.mark.parametrize("w,h,bands", [(1920, 1080, 8), (704, 520, 3)])
def test_me_row_bands_equal_full_picture(product, gpu_ctx, w, h, bands):
    """svt_amd_me_picture_range_launch over contiguous LCU-row bands (what each GPU of an LCU-row shard runs) leaves,
    band by band, exactly the records of the one-launch picture."""
    lib = product
    frames = [S.gen_luma("motion", w, h, t, 7) for t in range(3)]
    for s, f in enumerate(frames):
        upload(lib, gpu_ctx, s, f)
    p = default_params(w, h, num_lists=2, temporal_layer_index=1)
    full = me_picture(lib, gpu_ctx, p, 1, [0, 2])
    wl, hl = (w + 63) // 64, (h + 63) // 64
    nl = wl * hl
    refs = (C.c_int * 2)(0, 2)
    edges = [round(i * hl / bands) for i in range(bands + 1)]
    for b in reversed(range(bands)):  # any order: bands are independent
        if edges[b] == edges[b + 1]:
            continue
        rc = lib.svt_amd_me_picture_range_launch(gpu_ctx, C.byref(p), 1, refs, edges[b] * wl, edges[b + 1] * wl)
        assert rc == 0, lib.svt_amd_last_error()
    out = np.zeros(nl, S.ME_LCU_DTYPE)
    assert lib.svt_amd_me_picture_fetch(gpu_ctx, 1, out.ctypes.data) == 0, lib.svt_amd_last_error()
    assert np.array_equal(out["pu"], full["pu"])
    assert np.array_equal(out["best_sad"], full["best_sad"]) and np.array_equal(out["best_mv"], full["best_mv"])
    # a single band on its own, against the same rows of the full picture
    lo, hi = edges[1] * wl, edges[2] * wl
    upload(lib, gpu_ctx, 3, frames[1])
    rc = lib.svt_amd_me_picture_range_launch(gpu_ctx, C.byref(p), 3, refs, lo, hi)
    assert rc == 0, lib.svt_amd_last_error()
    one = np.zeros(nl, S.ME_LCU_DTYPE)
    assert lib.svt_amd_me_picture_fetch(gpu_ctx, 3, one.ctypes.data) == 0
    assert np.array_equal(one["pu"][lo:hi], full["pu"][lo:hi])


def test_three_lane_pipeline_batch_pack_and_lane_events(product, oracle):
    """the batched host pipeline bench.py runs (copy-in lane -> compute lane -> copy-out lane under lane events, records packed per
    batch into device arrays and moved with one copy each): the compact records that arrive in pinned host memory must be the
    corresponding parts of the blocking per-picture results, for two batches rotating through two buffer sets"""
    lib = product
    w, h, B = 448, 328, 3
    nl = S.lcu_count(w, h)
    vp = C.c_void_p
    root = vp()
    assert lib.svt_amd_context_create(0, 640, 384, 2 * B, C.byref(root)) == 0, lib.svt_amd_last_error()
    lanes = [vp(), vp(), vp()]
    try:
        for lane in lanes:
            assert lib.svt_amd_context_fork(root, C.byref(lane)) == 0, lib.svt_amd_last_error()
        lane_in, lane_k, lane_out = lanes
        p = default_params(w, h, num_lists=2, temporal_layer_index=1, cu8x8_mode=0)
        op = S.OisParams()
        op.luma_width, op.luma_height, op.ois_th_set, op.temporal_layer_index = w, h, 1, 1
        nc = lib.svt_amd_ois_compact_candidates(C.byref(op))
        me_b, ois_b = S.ME_PU_COUNT * 24, S.ME_PU_COUNT * nc * 4 + 88
        h_in = vp()
        assert lib.svt_amd_host_alloc(root, 2 * B * w * h, C.byref(h_in)) == 0
        frames = [S.gen_luma("motion", w, h, t, 9) for t in range(2 * B)]
        np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(2 * B * w * h,))[:] = np.concatenate([f.reshape(-1) for f in frames])
        sets = []
        for k in range(2):
            d_stage, d_me, d_ois, h_me, h_ois = vp(), vp(), vp(), vp(), vp()
            assert lib.svt_amd_device_alloc(lane_in, B * w * h, C.byref(d_stage)) == 0
            assert lib.svt_amd_device_alloc(lane_k, B * nl * me_b, C.byref(d_me)) == 0 and lib.svt_amd_device_alloc(lane_k, B * nl * ois_b, C.byref(d_ois)) == 0
            assert lib.svt_amd_host_alloc(lane_out, B * nl * me_b, C.byref(h_me)) == 0 and lib.svt_amd_host_alloc(lane_out, B * nl * ois_b, C.byref(h_ois)) == 0
            slots = (C.c_int * B)(*[k * B + i for i in range(B)])
            ptrs = (vp * B)(*[d_stage.value + i * w * h for i in range(B)])
            jobs, ojobs = (S.MeJob * B)(), (S.OisJob * B)()
            for i in range(B):
                jobs[i].params, jobs[i].cur_slot = p, slots[i]
                jobs[i].ref_slot[0], jobs[i].ref_slot[1] = slots[(i - 1) % B], slots[(i + 1) % B]
                ojobs[i].params, ojobs[i].cur_slot = op, slots[i]
            sets.append(dict(d_stage=d_stage, d_me=d_me, d_ois=d_ois, h_me=h_me, h_ois=h_ois, slots=slots, ptrs=ptrs, jobs=jobs, ojobs=ojobs))
        for rnd in range(2):          # the second round reuses both sets: the lane events must keep the stages apart
            for k, L in enumerate(sets):
                assert lib.svt_amd_lane_event_wait(lane_in, lane_k, k) == 0
                assert lib.svt_amd_device_upload_async(lane_in, L["d_stage"], vp(h_in.value + k * B * w * h), B * w * h) == 0
                assert lib.svt_amd_lane_event_record(lane_in, k) == 0
                assert lib.svt_amd_lane_event_wait(lane_k, lane_in, k) == 0 and lib.svt_amd_lane_event_wait(lane_k, lane_out, k) == 0
                assert lib.svt_amd_picture_upload_device_batch(lane_k, B, L["slots"], L["ptrs"], w, w, h) == 0, lib.svt_amd_last_error()
                assert lib.svt_amd_lane_event_record(lane_k, k) == 0
                assert lib.svt_amd_me_batch_launch(lane_k, L["jobs"], B) == 0, lib.svt_amd_last_error()
                assert lib.svt_amd_ois_batch_launch(lane_k, L["ojobs"], B) == 0, lib.svt_amd_last_error()
                assert lib.svt_amd_records_pack_batch_async(lane_k, L["slots"], B, nc, L["d_me"], L["d_ois"]) == 0, lib.svt_amd_last_error()
                assert lib.svt_amd_lane_event_record(lane_k, 2 + k) == 0 and lib.svt_amd_lane_event_wait(lane_out, lane_k, 2 + k) == 0
                assert lib.svt_amd_device_download_async(lane_out, L["h_me"], L["d_me"], B * nl * me_b) == 0
                assert lib.svt_amd_device_download_async(lane_out, L["h_ois"], L["d_ois"], B * nl * ois_b) == 0
                assert lib.svt_amd_lane_event_record(lane_out, k) == 0
        for lane in lanes:
            assert lib.svt_amd_synchronize(lane) == 0
        cdt = np.dtype([("candidate", "<u4", (S.ME_PU_COUNT, nc)), ("total", "u1", (S.ME_PU_COUNT,)), ("pad", "u1", (3,))])
        for k, L in enumerate(sets):
            cme = _records(L["h_me"], B * nl * S.ME_PU_COUNT, S.ME_LCU_DTYPE["pu"].base).reshape(B, nl, S.ME_PU_COUNT)
            cois = _records(L["h_ois"], B * nl, cdt).reshape(B, nl)
            for i in range(B):
                cur = k * B + i
                refs = [k * B + (i - 1) % B, k * B + (i + 1) % B]
                full = me_picture(lib, root, p, cur, refs)
                assert cme[i].tobytes() == np.ascontiguousarray(full["pu"]).tobytes(), (k, i)
                ois = np.zeros(nl, S.OIS_LCU_DTYPE)
                assert lib.svt_amd_ois_picture(root, C.byref(op), cur, None, ois.ctypes.data) == 0, lib.svt_amd_last_error()
                assert np.array_equal(cois[i]["candidate"], ois["candidate"][:, :, :nc]) and np.array_equal(cois[i]["total"], ois["total"]), (k, i)
        # and one picture against the oracle, so that "equal to the blocking calls" is not equal garbage
        pics = [S.OraclePicture(oracle, f) for f in frames[:B]]
        want = S.oracle_me_picture(oracle, p, pics[1], pics[0], pics[2])
        got = np.zeros(nl, S.ME_LCU_DTYPE)
        got["pu"] = _records(sets[0]["h_me"], B * nl * S.ME_PU_COUNT, S.ME_LCU_DTYPE["pu"].base).reshape(B, nl, S.ME_PU_COUNT)[1]
        assert np.array_equal(got["pu"], want["pu"])
    finally:
        for lane in lanes:
            if lane:
                lib.svt_amd_context_destroy(lane)
        lib.svt_amd_context_destroy(root)
