#!/usr/bin/env python3
"""GPU box, round 6: encoded fps of the bound encoder (library defaults = the closed loop) at BASELINE configs[2] over the launch width of the mode-decision kernel - since
the encode pass is a kernel of its own, a mode-decision workgroup no longer spends a quarter of its time encoding, and the grid that gave a picture its shortest kernel may be
narrower.  usage: fps_grid_sweep.py [frames] [wide:narrow ...]   (0:0 = the library's own choice)"""
import json
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_fps as E
import svtlib as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 160
specs = sys.argv[2:] or ["0:0", "40:16", "32:16", "32:12", "28:14", "40:24"]
w, h, depth, args = E.CONFIGS["cfg3"]
args = list(args) + ["-asm", "1", "-lp", "32"]
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    yuv = os.path.join(td, "clip.yuv")
    S.write_clip(yuv, "motion", w, h, 16, 7)
    ref = E.run_app(S.REF_APP, yuv, w, h, frames, args, os.path.join(td, "ref.265"), nb=16)
    print(json.dumps({"reference_fps": ref["fps"], "frames": frames}), flush=True)
    for spec in specs:
        wide, narrow = (int(v) for v in spec.split(":"))
        env = {}
        if wide:
            env["SVT_AMD_MD_WIDE"] = str(wide)
        if narrow:
            env["SVT_AMD_MD_NARROW"] = str(narrow)
        runs = []
        for _ in range(3):
            r = E.run_app(E.HIP_APP, yuv, w, h, frames, args, os.path.join(td, "hip.265"), env=env, nb=16)
            runs.append((r["fps"], r["md5"] == ref["md5"]))
        print(json.dumps({"wide": wide, "narrow": narrow, "fps": sorted(f for f, _ in runs), "identical": all(ok for _, ok in runs)}), flush=True)
