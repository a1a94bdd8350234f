#!/usr/bin/env python3
"""GPU box, a -DMD_TRACE build (tools/exp_build.sh trace -DMD_TRACE -DMD_INTER8_ONLY; SVT_PRODUCT_LIB=tools/_exp/lib_trace.so): lane-0 time stamps of the four waves of the
mode-decision kernel along two consecutive units of one LCU of a recorded 4K B picture - which wave the chain waits for at every barrier.
usage: md_trace.py [picture index] [lcu] [first unit]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S
from test_gpu_md import md_encode_inter, sig

N = 120
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
unit = int(sys.argv[3]) if len(sys.argv) > 3 else 6
g = dict(np.load(os.environ.get("MD_CHAIN_FX", os.path.join(ROOT, "tools", "_fx", "md4k.npz"))))
w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
lib = S.load_product()
sig(lib)
ctx, pic = C.c_void_p(), C.c_void_p()
assert lib.svt_amd_context_create(0, w, h, 2, C.byref(ctx)) == 0
assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0
lib.svt_amd_debug_md_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
md_encode_inter(lib, ctx, pic, g, k, encode=True)
lcus = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1050").split(",")]
for lcu in lcus:
    assert lib.svt_amd_debug_md_trace(ctx, pic, lcu, unit, None) == 0, "not a -DMD_TRACE build"
    md_encode_inter(lib, ctx, pic, g, k, encode=True)
    tr = np.zeros(4 * (1 + N) + 4, np.uint64)
    assert lib.svt_amd_debug_md_trace(ctx, pic, lcu, unit, tr.ctypes.data) == 0
    tr = tr[:4 * (1 + N)].reshape(4, 1 + N)
    L = g["lcu"][k][lcu]
    print("picture", k, "lcu", lcu, "leaf_count", int(L["leaf_count"]), "leaves", [int(v) for v in L["leaf_index"][:int(L["leaf_count"])]], "units from", unit)
    ev = []
    for wv in range(4):
        n = int(tr[wv, 0])
        for e in tr[wv, 1:1 + n]:
            ev.append((int(e) & 0xFFFFFFFFFFFF, wv, int(e) >> 48))
    ev.sort()
    t0 = ev[0][0] if ev else 0
    last = [t0] * 4
    for t, wv, m in ev:
        print("%8d  %s w%d m%-3d  (+%d)" % (t - t0, "        " * wv, wv, m, t - last[wv]))
        last[wv] = t
