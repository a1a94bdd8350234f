#!/usr/bin/env python3
"""GPU box: device-resident mode decision + encode pass of one I picture at a BASELINE size.  The unmodified reference (oracle/_ref, prebuilt)
encodes one picture with the recording harness on (SVT_REF_MD_DUMP); the device call runs on the recorded inputs, its decisions are compared
with the reference's, and the call is timed.  usage: md_bench.py [w h encMode reps]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import svtlib as S  # noqa: E402
from make_md_golden import parse_dump  # noqa: E402
from test_oracle_md_golden import compare_md  # noqa: E402


def record(w, h, enc_mode, kind="motion", extra=()):
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "c.yuv"), os.path.join(td, "md.dump")
        S.write_clip(yuv, kind, w, h, 1, 7)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", "1", "-asm", "1", "-b", os.path.join(td, "o.265"), "-encMode", str(enc_mode),
               "-intra-period", "0", "-q", "32"] + list(extra)
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_MD_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        pics, lcus = parse_dump(open(dump, "rb").read())
    h0, y, cb, cr, ois = pics[0][:5]
    r = lcus[lcus["picture_number"] == 0]
    r = r[np.argsort(r["lcu_index"])]
    return dict(pic=np.array([h0["pic"]]), cost=np.ascontiguousarray(h0["cost"]), y=np.ascontiguousarray(y), cb=np.ascontiguousarray(cb),
                cr=np.ascontiguousarray(cr), ois=np.ascontiguousarray(ois), lcu=np.ascontiguousarray(r["lcu"]), out=np.ascontiguousarray(r["out"]))


def run(lib, g, reps=5, check=True):
    from test_gpu_md import sig
    sig(lib)
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    n = len(g["lcu"])
    out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)
    ts = []
    prof = os.environ.get("MD_BENCH_PROFILE")
    if prof:
        lib.svt_amd_debug_md_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert lib.svt_amd_debug_md_profile(ctx, pic, None) == 0
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rc = lib.svt_amd_md_encode_picture(ctx, pic, g["pic"].ctypes.data, g["lcu"].ctypes.data, g["y"].ctypes.data, w, g["cb"].ctypes.data, g["cr"].ctypes.data,
                                           w // 2, g["ois"].ctypes.data, 0, g["cost"].ctypes.data, out.ctypes.data, works.ctypes.data, res.ctypes.data)
        assert rc == 0, lib.svt_amd_last_error()
        ts.append((time.perf_counter() - t0) * 1e3)
    stages = None
    if prof:
        pr = np.zeros((n, 16), np.uint64)
        assert lib.svt_amd_debug_md_profile(ctx, pic, pr.ctypes.data) == 0
        names = ["load", "lane0_candidates", "intra_ref", "fast_loop", "lane0_fast_costs", "full_loop", "lane0_decision", "recon_interdepth", "update_next",
                 "store", "work_record", "encode_pass", "wait_neighbours"]
        tot = pr[:, :13].sum(axis=0).astype(np.float64)
        calls = float(pr[:, 15].sum()) / n
        stages = {nm: round(float(tot[i]) / n / calls, 0) for i, nm in enumerate(names)}   # shader clocks per LCU per call
        stages["sum_without_wait"] = round(float(tot[:12].sum()) / n / calls, 0)
        stages["units_tested_per_lcu"] = round(float(pr[:, 14].sum()) / n / calls, 2)
        stages["fast_loop_candidates_per_unit"] = round(float(pr[:, 13].sum()) / max(1.0, float(pr[:, 14].sum())), 2)
        sub = np.zeros((n, 16), np.uint64)
        lib.svt_amd_debug_md_profile_sub.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if lib.svt_amd_debug_md_profile_sub(ctx, pic, sub.ctypes.data) == 0:   # the finer marks (MD_SUB), shader clocks per tested unit
            stages["sub_stage_clocks_per_unit"] = [round(float(v) / max(1.0, float(pr[:, 14].sum())), 0) for v in sub.sum(axis=0)]
    if check:
        compare_md(out, g["out"], "%dx%d" % (w, h))
    lib.svt_amd_encdec_picture_destroy(ctx, pic)
    lib.svt_amd_context_destroy(ctx)
    units = int(works["num_cus"].sum())
    tested = int(g["out"]["tested"].sum())
    return {"stage_clocks_per_lcu": stages, "width": w, "height": h, "lcus": n, "leaves_tested": tested, "final_units": units, "ms_first_call": round(ts[0], 2),
            "ms_per_picture": round(float(np.median(ts[1:])), 2), "pictures_per_s_one_in_flight": round(1e3 / float(np.median(ts[1:])), 2),
            "what": "svt_amd_md_encode_picture through the host-array ABI (source planes, OIS and LCU controls up, decisions + work + result records down), "
                    "decisions identical to the reference's ModeDecisionLcu records of the same picture"}


def record_inter(w, h, enc_mode, frames=9, kind="objects", levels=3, extra=(), ref=False):
    """a random-access encode of `frames` pictures by the prebuilt reference with the recording harness on; returns the fixture-shaped records of its
    open-loop P / B pictures whose LCUs all went through ModeDecisionLcu (the pictures svt_amd_md_encode_picture_inter covers): ref = False the
    non-reference ones (luma-only candidates), True the reference ones (chroma level 4: CHROMA_MODE_FULL LCUs), None both"""
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "c.yuv"), os.path.join(td, "md.dump")
        S.write_clip(yuv, kind, w, h, frames, 7)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(frames), "-asm", "1", "-b", os.path.join(td, "o.265"), "-encMode", str(enc_mode),
               "-pred-struct", "2", "-hierarchical-levels", str(levels), "-q", "32"] + list(extra)
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_MD_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        pics, lcus = parse_dump(open(dump, "rb").read())
    nl = S.lcu_count(w, h)
    keep = [p for p in sorted(pics) if pics[p][0]["pic"]["slice_type"] != 2 and pics[p][0]["pic"]["intra_md_open_loop"] and (lcus["picture_number"] == p).sum() == nl and
            (ref is None or bool(pics[p][0]["pic"]["is_reference"]) == bool(ref))]
    g = {"picture_number": np.array(keep, np.uint64)}
    recs = []
    for p in keep:
        r = lcus[lcus["picture_number"] == p]
        recs.append(r[np.argsort(r["lcu_index"])])
    g["pic"] = np.stack([pics[p][0]["pic"] for p in keep])
    g["cost"] = np.stack([pics[p][0]["cost"] for p in keep])
    for i, nm in enumerate(("src_y", "src_cb", "src_cr", "ois")):
        g[nm] = np.stack([pics[p][1 + i] for p in keep])
    g["lcu"] = np.stack([r["lcu"] for r in recs])
    g["out"] = np.stack([r["out"] for r in recs])
    g["inter"] = np.stack([pics[p][0]["inter"] for p in keep])
    g["me"] = np.stack([pics[p][5][0] for p in keep])["pu"]
    g["tmvp_present"] = np.array([pics[p][5][1] is not None for p in keep])
    g["tmvp"] = np.stack([pics[p][5][1] if pics[p][5][1] is not None else np.zeros(nl, S.MD_TMVP_LCU_DTYPE) for p in keep])
    g["ref_geom"] = np.stack([np.array([pics[p][0][k] for k in ("ref_stride_y", "ref_stride_c", "ref_origin_x", "ref_origin_y", "ref_width", "ref_height", "nref")],
                                       np.uint32) for p in keep])
    for l in range(2):
        for k, nm in enumerate(("y", "cb", "cr")):
            g["ref%d_%s" % (l, nm)] = np.stack([pics[p][5][2][min(l, len(pics[p][5][2]) - 1)][k] for p in keep])

    class G(dict):
        files = property(lambda self: list(self.keys()))
    return G(g)


def run_inter(lib, g, reps=3, encode=True, check=True, profile=None):
    """svt_amd_md_encode_picture_inter on every recorded picture: decisions vs the reference's, time per call"""
    from test_gpu_md import sig, md_encode_inter
    sig(lib)
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    prof = os.environ.get("MD_BENCH_PROFILE") if profile is None else profile
    if prof:
        lib.svt_amd_debug_md_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert lib.svt_amd_debug_md_profile(ctx, pic, None) == 0
    lib.svt_amd_debug_md_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    ts, tested, units, kms, eps = [], 0, 0, [[] for _ in g["picture_number"]], [[] for _ in g["picture_number"]]
    wgs = C.c_int(0)
    has_ep = hasattr(lib, "svt_amd_debug_md_ep_ms")
    if has_ep:
        lib.svt_amd_debug_md_ep_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    for k in range(len(g["picture_number"])):
        for rep in range(reps):
            t0 = time.perf_counter()
            out, works, _ = md_encode_inter(lib, ctx, pic, g, k, encode=encode)
            if rep:
                ts.append((time.perf_counter() - t0) * 1e3)
                ms = C.c_float(0)
                assert lib.svt_amd_debug_md_kernel_ms(ctx, pic, C.byref(ms), C.byref(wgs)) == 0, lib.svt_amd_last_error()
                kms[k].append(float(ms.value))
                if has_ep and encode:
                    e = C.c_float(0)
                    assert lib.svt_amd_debug_md_ep_ms(ctx, pic, C.byref(e)) == 0, lib.svt_amd_last_error()
                    eps[k].append(float(e.value))
        if check:
            compare_md(out, g["out"][k], "%dx%d picture %d" % (w, h, int(g["picture_number"][k])))
        tested += int(g["out"][k]["tested"].sum())
        units += int(works["num_cus"].sum())
    n = g["lcu"].shape[1]
    stages = None
    if prof:
        pr = np.zeros((n, 16), np.uint64)
        assert lib.svt_amd_debug_md_profile(ctx, pic, pr.ctypes.data) == 0
        names = ["load", "lane0_candidates", "intra_ref", "fast_loop", "lane0_fast_costs", "full_loop", "lane0_decision", "recon_interdepth", "update_next",
                 "store", "work_record", "encode_pass", "wait_neighbours"]
        tot = pr[:, :13].sum(axis=0).astype(np.float64)
        calls = float(pr[:, 15].sum()) / n
        stages = {nm: round(float(tot[i]) / n / calls, 0) for i, nm in enumerate(names)}
        stages["sum_without_wait"] = round(float(tot[:12].sum()) / n / calls, 0)
        stages["fast_loop_candidates_per_unit"] = round(float(pr[:, 13].sum()) / max(1.0, float(pr[:, 14].sum())), 2)
        stages["units_tested_per_lcu"] = round(float(pr[:, 14].sum()) / n / calls, 2)
        if os.environ.get("MD_BENCH_RAW"):
            stages["raw_sums"] = [int(v) for v in pr.sum(axis=0)]
    lib.svt_amd_encdec_picture_destroy(ctx, pic)
    lib.svt_amd_context_destroy(ctx)
    per_pic = [{"picture": int(p), "temporal_layer": int(g["pic"][k]["temporal_layer"]), "is_reference": int(g["pic"][k]["is_reference"]),
                "chroma_level": int(g["pic"][k]["chroma_level"]), "kernel_ms": round(float(np.median(kms[k])), 3) if kms[k] else None,
                "encode_pass_kernel_ms": round(float(np.median(eps[k])), 3) if eps[k] else None,
                "leaves_tested": int(g["out"][k]["tested"].sum())} for k, p in enumerate(g["picture_number"])]
    return {"kernel": {"name": "k_md_picture<true> (+ k_encode_picture behind it on the same stream)", "workgroups": int(wgs.value), "ms_by_hip_events": per_pic},
            "stage_clocks_per_lcu": stages, "width": w, "height": h, "lcus": n, "pictures": [int(p) for p in g["picture_number"]], "leaves_tested_per_picture": tested // len(ts) * (reps - 1) if ts else 0,
            "final_units": units, "ms_per_picture_incl_host_copies": round(float(np.median(ts)), 2),
            "what": "svt_amd_md_encode_picture_inter (mode decision%s) of the recorded B pictures through the host-array ABI incl. reference-picture upload by the test; "
                    "decisions identical to the reference's ModeDecisionLcu records" % (" + merge / skip decision + encode pass" if encode else " only")}


def run_inter_flights(lib, g, flights=2, reps=4):
    """`flights` pictures in flight: one thread, lane (forked context) and picture object per flight, every thread repeating the device call of picture 0.  Everything
    a call needs is staged ONCE per flight (reference pictures resident and set, records and output arrays allocated): inside the timed loop a thread only makes the
    C call - ctypes releases the interpreter lock for its duration, so the threads really overlap (with the test helper in the loop, which re-uploads the reference
    pictures through torch and allocates 55 MB of numpy output per call under the lock, eight threads measure the interpreter, not the device)."""
    import threading
    import torch
    from test_gpu_md import sig
    from test_oracle_md_golden import inter_inputs
    sig(lib)
    vp = C.c_void_p
    lib.svt_amd_md_encode_picture_inter.restype = C.c_int
    lib.svt_amd_md_encode_picture_inter.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    lib.svt_amd_encdec_picture_set_inter.restype = C.c_int
    lib.svt_amd_encdec_picture_set_inter.argtypes = [vp] * 5
    lib.svt_amd_debug_md_kernel_ms.restype = C.c_int
    lib.svt_amd_debug_md_kernel_ms.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    root = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(root)) == 0, lib.svt_amd_last_error()
    lib.svt_amd_context_fork.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    k = 0
    P = np.ascontiguousarray(g["pic"][k:k + 1])
    lcus = np.ascontiguousarray(g["lcu"][k])
    cost = np.ascontiguousarray(g["cost"][k])
    src = [np.ascontiguousarray(g[n][k]) for n in ("src_y", "src_cb", "src_cr")]
    o = np.ascontiguousarray(g["ois"][k])
    X, me, tmvp, refs, planes = inter_inputs(g, k)
    dev = [[torch.from_numpy(a).cuda() for a in pl] for pl in planes]
    torch.cuda.synchronize()
    rs = [S.RefPicture(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), r.strideY, r.strideC, r.originX, r.originY, r.width, r.height) for d, r in zip(dev, refs)]
    n = len(lcus)
    lanes, pics, outs = [], [], []
    for i in range(flights):
        lane, pic = C.c_void_p(), C.c_void_p()
        assert lib.svt_amd_context_fork(root, C.byref(lane)) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_encdec_picture_create(lane, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
        assert lib.svt_amd_encdec_picture_set_inter(lane, pic, C.byref(rs[0]), C.byref(rs[1]), cost.ctypes.data) == 0, lib.svt_amd_last_error()
        lanes.append(lane), pics.append(pic)
        outs.append((np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)))

    def call(i):
        out, works, res = outs[i]
        rc = lib.svt_amd_md_encode_picture_inter(lanes[i], pics[i], P.ctypes.data, X.ctypes.data, lcus.ctypes.data, src[0].ctypes.data, src[0].shape[1], src[1].ctypes.data,
                                                 src[2].ctypes.data, src[1].shape[1], o.ctypes.data, 0, me.ctypes.data, 0, tmvp.ctypes.data if tmvp is not None else None,
                                                 out.ctypes.data, works.ctypes.data, res.ctypes.data)
        assert rc == 0, lib.svt_amd_last_error()
    for i in range(flights):
        call(i)   # warm-up: allocations
    times, kms = [[] for _ in range(flights)], [[] for _ in range(flights)]

    def work(i):
        for _ in range(reps):
            t0 = time.perf_counter()
            call(i)
            times[i].append((time.perf_counter() - t0) * 1e3)
            ms = C.c_float()
            if lib.svt_amd_debug_md_kernel_ms(lanes[i], pics[i], C.byref(ms), None) == 0:
                kms[i].append(ms.value)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(flights)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    for lane, pic in zip(lanes, pics):
        lib.svt_amd_encdec_picture_destroy(lane, pic)
        lib.svt_amd_context_destroy(lane)
    lib.svt_amd_context_destroy(root)
    del dev
    return {"flights": flights, "ms_per_call_median": round(float(np.median(np.concatenate(times))), 2), "kernel_ms_median": round(float(np.median(np.concatenate(kms))), 2),
            "pictures_per_s": round(flights * reps / wall, 2)}


if __name__ == "__main__":
    a = sys.argv[1:]
    w, h, m, reps = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (3840, 2160, 7, 5)
    if len(a) >= 5 and a[4] == "inter":
        g = record_inter(w, h, m, frames=int(a[5]) if len(a) > 5 else 9, kind=os.environ.get("MD_BENCH_CLIP", "objects"), levels=2,
                         ref={"ref": True, "all": None}.get(os.environ.get("MD_BENCH_PICTURES", ""), False))
        lib = S.load_product()
        out = run_inter(lib, g, reps)
        if os.environ.get("MD_BENCH_FLIGHTS"):
            out["in_flight"] = [run_inter_flights(lib, g, f) for f in (int(v) for v in os.environ["MD_BENCH_FLIGHTS"].split(","))]
        print(json.dumps(out))
    else:
        g = record(w, h, m)
        print(json.dumps(run(S.load_product(), g, reps)))
