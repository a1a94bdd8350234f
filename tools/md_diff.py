#!/usr/bin/env python3
"""GPU box: decisions of the device call against the reference's records for one picture of a recorded fixture; prints the differing leaves of the first differing LCUs.
usage: md_diff.py <fixture.npz> <picture index>"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S
from test_gpu_md import md_encode_inter, sig
g = dict(np.load(sys.argv[1]))
k = int(sys.argv[2])
w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
lib = S.load_product()
sig(lib)
ctx, pic = C.c_void_p(), C.c_void_p()
assert lib.svt_amd_context_create(0, w, h, 2, C.byref(ctx)) == 0
assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0
out, _, _ = md_encode_inter(lib, ctx, pic, g, k, encode=True)
want = g["out"][k]
nbad = 0
for i in range(len(want)):
    t = want[i]["tested"] == 1
    bad = [l for l in np.nonzero(t)[0] if any(not np.array_equal(out[i][f][l], want[i][f][l]) for f in ("split", "pred_mode", "cost", "mv", "merge_flag", "inter_dir", "intra_luma_mode"))]
    if not np.array_equal(out[i]["tested"], want[i]["tested"]):
        print("LCU", i, "tested sets differ")
    if bad:
        nbad += 1
        if nbad <= 4:
            L = g["lcu"][k][i]
            print("LCU", i, "leaves", [int(v) for v in L["leaf_index"][:int(L["leaf_count"])]], "chroma_encode_mode", int(L["chroma_encode_mode"]), "lcu_md_mode", int(L["lcu_md_mode"]))
            for l in bad[:3]:
                for nm, r in (("got ", out[i]), ("want", want[i])):
                    print("   leaf", l, nm, {f: r[f][l].tolist() for f in ("split", "pred_mode", "intra_luma_mode", "ycbf", "inter_dir", "merge_flag", "merge_index", "mv", "cost", "merge_cost", "skip_cost")})
print("LCUs with differing leaves:", nbad, "of", len(want))
