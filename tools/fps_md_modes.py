#!/usr/bin/env python3
"""GPU box: encoded fps at BASELINE configs[2] of the reference and of the bound encoder with the device's mode decision for (a) every covered picture (SVT_HOOK_MD=1)
and (b) P / B pictures only (SVT_HOOK_MD=pb: I pictures stay with the reference code), for several clip lengths - a 4K I picture's closed-loop decision takes the device
0.45 s and everything waits for it, so short clips measure the I pictures.  usage: fps_md_modes.py [lp] [frames,...]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import encoder_fps as E

lp = sys.argv[1] if len(sys.argv) > 1 else "32"
for frames in (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64,160").split(",")):
    row = {"lp": int(lp), "frames": frames}
    only = os.environ.get("FPS_MODES", "front_half_only,md_all,md_pb").split(",")
    for tag, env in (("front_half_only", {}), ("md_all", {"SVT_HOOK_MD": "1"}), ("md_pb", {"SVT_HOOK_MD": "pb"})):
        if tag not in only:
            continue
        r = E.measure("cfg3", frames=frames, extra=["-lp", lp], hip_env=env, unique=16)
        row["reference_fps"] = r["reference"]["fps"]
        row[tag + "_fps"] = r["hip"]["fps"]
        row[tag + "_identical"] = r["bitstream_identical"]
    print(json.dumps(row), flush=True)
