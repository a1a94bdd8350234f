#!/usr/bin/env python3
"""GPU box: N encodes of BASELINE configs[2] (160 pictures) by the bound encoder with the library's defaults (the closed loop bench.py measures) under the watchdog
(SVT_HOOK_WATCHDOG=20: an encoder without an LCU through EncodePass for 20 s ends itself and says what every picture object and the launch budget held) - round 5 saw ONE
such wedge that never reproduced (profiles/r05_o_stress_watchdog.txt); round 6 changed the device call (two kernels, no encode pass inside the persistent kernel).
usage: stress_closed_loop.py [runs] [frames]"""
import json
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_fps as E
import svtlib as S

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 160
w, h, depth, args = E.CONFIGS["cfg3"]
args = list(args) + ["-asm", "1", "-lp", "32"]
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    yuv = os.path.join(td, "clip.yuv")
    S.write_clip(yuv, "motion", w, h, 16, 7)
    ref = E.run_app(S.REF_APP, yuv, w, h, frames, args, os.path.join(td, "ref.265"), nb=16)
    print(json.dumps({"reference_fps": ref["fps"], "md5": ref["md5"], "frames": frames}), flush=True)
    fps, bad, aborted = [], 0, []
    t0 = time.time()
    for k in range(runs):
        try:
            r = E.run_app(E.HIP_APP, yuv, w, h, frames, args, os.path.join(td, "hip.265"), env={"SVT_HOOK_WATCHDOG": "20"}, nb=16, timeout=120)
            fps.append(r["fps"])
            bad += r["md5"] != ref["md5"]
        except Exception as e:
            aborted.append({"run": k, "what": str(e)[-1500:]})
            print(json.dumps(aborted[-1]), flush=True)
    fps.sort()
    print(json.dumps({"runs": runs, "completed": len(fps), "aborted_or_timed_out": len(aborted), "bitstreams_differing": bad, "fps_min": fps[0] if fps else None,
                      "fps_median": fps[len(fps) // 2] if fps else None, "fps_max": fps[-1] if fps else None, "wall_s": round(time.time() - t0, 1)}), flush=True)
