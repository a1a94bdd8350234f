#!/usr/bin/env python3
"""GPU box: where the mode-decision kernel's time goes ALONG THE WAVEFRONT, by stage and by unit depth (round 6's working tool; md_chain.py is round 5's).
Recorded 4K B pictures of BASELINE configs[2] (tools/_fx/md4k.npz, recorded once with the prebuilt reference by tools/md_record_4k.py - here or on the box); per picture: a
warm-up call, a timed call, a profiled call (svt_amd_debug_md_profile*), decisions compared with the reference's records (compare_md), the longest dependency path of the
wavefront from the per-LCU clocks and the stage / sub-stage clocks per unit on that path, split by the unit's depth.
usage: md_chain2.py [picture indices, default "0,1"]   env: SVT_PRODUCT_LIB=<experimental build>, MD_CHAIN_FX=<fixture>"""
import ctypes as C
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S
from test_gpu_md import md_encode_inter, sig
from test_oracle_md_golden import compare_md

STAGES = ["load", "lane0_candidates", "intra_ref", "fast_loop", "lane0_fast_costs", "full_loop", "lane0_decision", "recon_interdepth", "update_next", "store"]


def main():
    fx = os.environ.get("MD_CHAIN_FX", os.path.join(ROOT, "tools", "_fx", "md4k.npz"))
    if not os.path.exists(fx):
        import md_record_4k
        md_record_4k.record(fx)
    g = dict(np.load(fx))
    which = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
    mhz = 2400.0
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    lib = S.load_product()
    sig(lib)
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, h, 2, C.byref(ctx)) == 0
    lib.svt_amd_debug_md_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    for f in ("svt_amd_debug_md_profile", "svt_amd_debug_md_profile_sub", "svt_amd_debug_md_profile_depth"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    res = {"lib": os.environ.get("SVT_PRODUCT_LIB", "product"), "pictures": []}
    for k in which:
        pic = C.c_void_p()
        assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0
        md_encode_inter(lib, ctx, pic, g, k, encode=True)          # warm-up (allocations)
        ms_runs, ep_ms = [], None
        for _ in range(3):
            out, works, _r = md_encode_inter(lib, ctx, pic, g, k, encode=True)
            ms0, wg = C.c_float(0), C.c_int(0)
            lib.svt_amd_debug_md_kernel_ms(ctx, pic, C.byref(ms0), C.byref(wg))
            ms_runs.append(round(float(ms0.value), 2))
            if hasattr(lib, "svt_amd_debug_md_ep_ms"):
                lib.svt_amd_debug_md_ep_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
                e0 = C.c_float(0)
                lib.svt_amd_debug_md_ep_ms(ctx, pic, C.byref(e0))
                ep_ms = round(float(e0.value), 2)
        ok = True
        try:
            compare_md(out, g["out"][k], "picture %d" % k)
        except AssertionError as e:
            ok = False
            print("DECISIONS DIFFER:", str(e)[:400], file=sys.stderr)
        assert lib.svt_amd_debug_md_profile(ctx, pic, None) == 0
        md_encode_inter(lib, ctx, pic, g, k, encode=True)          # the profiled call
        ms = C.c_float(0)
        lib.svt_amd_debug_md_kernel_ms(ctx, pic, C.byref(ms), None)
        n = g["lcu"].shape[1]
        pr = np.zeros((n, 16), np.uint64)
        assert lib.svt_amd_debug_md_profile(ctx, pic, pr.ctypes.data) == 0
        pr = pr.astype(np.float64)
        wl, hl = (w + 63) // 64, (h + 63) // 64
        t_md = pr[:, 0:10].sum(axis=1)
        units = pr[:, 14]
        T = np.zeros(n)
        pred = np.full(n, -1)
        for y in range(hl):
            for x in range(wl):
                i = y * wl + x
                d0 = i - 1 if x > 0 else -1
                d1 = -1 if y == 0 else (i - wl + 1 if x + 1 < wl else i - wl)
                s = 0.0
                for d in (d0, d1):
                    if d >= 0 and T[d] > s:
                        s, pred[i] = T[d], d
                T[i] = s + t_md[i]
        i = int(np.argmax(T))
        path = []
        while i >= 0:
            path.append(i)
            i = int(pred[i])
        path = path[::-1]
        o = {"picture": int(g["picture_number"][k]), "temporal_layer": int(g["pic"][k]["temporal_layer"]), "is_reference": int(g["pic"][k]["is_reference"]),
             "decisions_identical": ok, "kernel_ms": ms_runs, "ep_kernel_ms": ep_ms, "kernel_ms_profiled": round(float(ms.value), 2), "workgroups": int(wg.value),
             "longest_path_ms": round(float(T.max()) / (mhz * 1e3), 2), "units_on_path": int(units[path].sum()), "units_per_lcu_on_path": round(float(units[path].mean()), 2),
             "clocks_per_unit_on_path": round(float(t_md[path].sum() / units[path].sum()), 0),
             "stage_clocks_per_unit_on_path": {nm: round(float(pr[path, j].sum() / units[path].sum()), 0) for j, nm in enumerate(STAGES)},
             "ep_ms_per_lcu_mean": round(float((pr[:, 10] + pr[:, 11]).mean()) / (mhz * 1e3), 4)}
        sub = np.zeros((n, 16), np.uint64)
        if lib.svt_amd_debug_md_profile_sub(ctx, pic, sub.ctypes.data) == 0:
            o["sub_stage_clocks_per_unit_on_path"] = [round(float(sub[path, j].astype(np.float64).sum() / units[path].sum()), 0) for j in range(16)]
        dp = np.zeros((n, 4, 32), np.uint64)
        if hasattr(lib, "svt_amd_debug_md_profile_depth") and lib.svt_amd_debug_md_profile_depth(ctx, pic, dp.ctypes.data) == 0:
            dp = dp.astype(np.float64)[path].sum(axis=0)   # [depth][32] over the LCUs on the path
            by = {}
            for d in range(4):
                nu = dp[d][14]
                if nu <= 0:
                    continue
                by["depth%d_%dx%d" % (d, 64 >> d, 64 >> d)] = {
                    "units": int(nu), "share_of_path": round(float(dp[d][0:10].sum() / t_md[path].sum()), 3), "clocks_per_unit": round(float(dp[d][0:10].sum() / nu), 0),
                    "candidates_per_unit": round(float(dp[d][13] / nu), 2),
                    "stages": {nm: round(float(dp[d][j] / nu), 0) for j, nm in enumerate(STAGES) if dp[d][j] > 0},
                    "sub": [round(float(dp[d][16 + j] / nu), 0) for j in range(16)]}
            o["by_depth_on_path"] = by
        res["pictures"].append(o)
        lib.svt_amd_encdec_picture_destroy(ctx, pic)
    lib.svt_amd_context_destroy(ctx)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
