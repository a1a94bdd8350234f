"""How the closed loop on the device scales with the pictures the encoder keeps in flight: the hooked encoder with SVT_HOOK_MD=1 (mode decision + encode pass of every
covered picture ONE device call) over -lp, which sizes the reference's picture-control-set pool (pictureControlSetPoolInitCountChild = max(4, lp / 6),
Source/Lib/Codec/EbEncHandle.c:1801) and its thread counts.  The reference runs once per -lp too (its own best is the bar).  md5-gated.
usage (GPU box): python tools/fps_md_lp.py [frames] [lp ...]      env: FPS_MD_ENV="K=V K=V" extra switches for the hooked run"""
import json
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import encoder_fps as E
import svtlib as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 96
lps = [int(v) for v in sys.argv[2:]] or [32, 64, 96, 128]
w, h, depth, args = E.CONFIGS["cfg3"]
extra_env = dict(kv.split("=", 1) for kv in os.environ.get("FPS_MD_ENV", "").split() if "=" in kv)
out = []
with tempfile.TemporaryDirectory() as td:
    yuv = os.path.join(td, "clip.yuv")
    S.write_clip(yuv, "motion", w, h, 16, 7)
    for lp in lps:
        a = list(args) + ["-asm", "1", "-lp", str(lp)]
        ref = E.run_app(S.REF_APP, yuv, w, h, frames, a, os.path.join(td, "ref.265"), nb=16)
        rp = os.path.join(td, "report_%d.txt" % lp)
        hip = E.run_app(E.HIP_APP, yuv, w, h, frames, a, os.path.join(td, "hip.265"), env=dict({"SVT_HOOK_MD": "1", "SVT_HOOK_REPORT": rp}, **extra_env), nb=16)
        front = E.run_app(E.HIP_APP, yuv, w, h, frames, a, os.path.join(td, "hip0.265"), nb=16)
        rep = [l for l in open(rp).read().splitlines() if "mode decision" in l]
        out.append({"lp": lp, "reference_fps": ref["fps"], "hip_front_half_only_fps": front["fps"], "hip_md_fps": hip["fps"], "md_identical": hip["md5"] == ref["md5"],
                    "front_identical": front["md5"] == ref["md5"], "report": rep})
        print(json.dumps(out[-1]), flush=True)
print(json.dumps(out))
