"""What bounds the encoder's fps at BASELINE configs[2]: the reference and the hooked encoder over the number of logical processors the library may use (-lp),
the hooked encoder also with the P / B closed loop on the device (SVT_HOOK_MD=pb).
usage (GPU box): python tools/fps_threads.py [frames] [lp ...]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import encoder_fps as E

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 96
lps = [int(v) for v in sys.argv[2:]] or [32, 64, 128, 256]
out = []
for lp in lps:
    r = E.measure("cfg3", frames=frames, unique=16, extra=["-lp", str(lp)])
    m = E.measure("cfg3", frames=frames, unique=16, extra=["-lp", str(lp)], hip_env={"SVT_HOOK_MD": "pb"})
    out.append({"lp": lp, "reference_fps": r["reference"]["fps"], "hip_fps": r["hip"]["fps"], "identical": r["bitstream_identical"],
                "hip_md_pb_fps": m["hip"]["fps"], "md_pb_identical": m["bitstream_identical"], "reference_fps_2nd_run": m["reference"]["fps"]})
    print(out[-1], flush=True)
print(json.dumps(out))
