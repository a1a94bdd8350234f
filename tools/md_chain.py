#!/usr/bin/env python3
"""GPU box: where the mode-decision kernel's time goes ALONG THE WAVEFRONT.  One recorded 4K B picture of BASELINE configs[2]; one profiled device call (stage clocks per LCU,
svt_amd_debug_md_profile); from the per-LCU mode-decision times the longest dependency path of the wavefront (an LCU starts when its left and its top-right neighbour are done:
md_done flags) is computed on the host and compared with the kernel's measured duration.  usage: md_chain.py [ref|nonref] [clock_mhz]"""
import ctypes as C
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import md_bench
import svtlib as S
from test_gpu_md import md_encode_inter, sig

ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2400.0   # __builtin_readcyclecounter = s_memtime: shader clocks (the longest path equals the kernel time at ~2.4 GHz)
w, h = 3840, 2160
g = md_bench.record_inter(w, h, 7, frames=5, kind="motion", levels=2, ref=ref)
lib = S.load_product()
sig(lib)
ctx, pic = C.c_void_p(), C.c_void_p()
assert lib.svt_amd_context_create(0, w, h, 2, C.byref(ctx)) == 0
assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0
lib.svt_amd_debug_md_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
lib.svt_amd_debug_md_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
md_encode_inter(lib, ctx, pic, g, 0, encode=True)          # warm-up (allocations)
ms0 = C.c_float(0)
md_encode_inter(lib, ctx, pic, g, 0, encode=True)
lib.svt_amd_debug_md_kernel_ms(ctx, pic, C.byref(ms0), None)
assert lib.svt_amd_debug_md_profile(ctx, pic, None) == 0
md_encode_inter(lib, ctx, pic, g, 0, encode=True)          # the profiled call
ms = C.c_float(0)
lib.svt_amd_debug_md_kernel_ms(ctx, pic, C.byref(ms), None)
n = g["lcu"].shape[1]
pr = np.zeros((n, 16), np.uint64)
assert lib.svt_amd_debug_md_profile(ctx, pic, pr.ctypes.data) == 0
pr = pr.astype(np.float64)
wl, hl = (w + 63) // 64, (h + 63) // 64
t_md = pr[:, 0:10].sum(axis=1)                 # load .. store: the LCU's mode decision (what the next LCUs wait for)
t_ep = pr[:, 10] + pr[:, 11]                   # work record + merge / skip decisions, encode pass
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
end = int(np.argmax(T))
path = []
i = end
while i >= 0:
    path.append(i)
    i = int(pred[i])
path = path[::-1]
to_ms = lambda c: c / (mhz * 1e3)
out = {"picture": int(g["picture_number"][0]), "reference_picture": bool(ref), "kernel_ms_unprofiled": round(float(ms0.value), 2), "kernel_ms_profiled": round(float(ms.value), 2),
       "lcus": n, "clock_mhz_assumed": mhz,
       "md_ms_per_lcu": {"mean": round(to_ms(t_md.mean()), 4), "median": round(to_ms(np.median(t_md)), 4), "p90": round(to_ms(np.percentile(t_md, 90)), 4), "max": round(to_ms(t_md.max()), 4)},
       "ep_ms_per_lcu_mean": round(to_ms(t_ep.mean()), 4), "units_per_lcu": {"mean": round(float(units.mean()), 2), "max": int(units.max())},
       "wavefront": {"steps_min": wl + 2 * (hl - 1), "mean_lcu_x_steps_ms": round(to_ms(t_md.mean()) * (wl + 2 * (hl - 1)), 2),
                     "longest_dependency_path_ms": round(to_ms(T.max()), 2), "lcus_on_it": len(path),
                     "mean_md_ms_of_lcus_on_it": round(to_ms(t_md[path].mean()), 4), "mean_units_of_lcus_on_it": round(float(units[path].mean()), 2)},
       "sum_of_all_lcus_md_ms": round(to_ms(t_md.sum()), 1), "sum_of_all_lcus_ep_ms": round(to_ms(t_ep.sum()), 1),
       "wait_for_neighbours_ms_per_lcu_mean": round(to_ms(pr[:, 12].mean()), 4),
       "stage_share_on_the_path": {nm: round(float(pr[path, k].sum() / t_md[path].sum()), 3) for k, nm in enumerate(
           ["load", "lane0_candidates", "intra_ref", "fast_loop", "lane0_fast_costs", "full_loop", "lane0_decision", "recon_interdepth", "update_next", "store"])}}
sub = np.zeros((n, 16), np.uint64)
lib.svt_amd_debug_md_profile_sub.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
if lib.svt_amd_debug_md_profile_sub(ctx, pic, sub.ctypes.data) == 0:
    sub = sub.astype(np.float64)
    out["sub_stage_clocks_per_unit_on_the_path"] = [round(float(sub[path, k].sum() / units[path].sum()), 0) for k in range(16)]
    out["stage_clocks_per_unit_on_the_path"] = {nm: round(float(pr[path, k].sum() / units[path].sum()), 0) for k, nm in enumerate(
        ["load", "lane0_candidates", "intra_ref", "fast_loop", "lane0_fast_costs", "full_loop", "lane0_decision", "recon_interdepth", "update_next", "store"])}
print(json.dumps(out))
if os.environ.get("MD_CHAIN_DUMP"):
    np.save(os.environ["MD_CHAIN_DUMP"], pr)
