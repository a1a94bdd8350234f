#!/usr/bin/env python3
"""GPU box: encoded fps at BASELINE configs[2] with the closed loop on the device (SVT_HOOK_MD) over the size of the EncDec picture pool (SVT_HOOK_PCS_POOL,
integration/svt_hook_encdec.c: the reference's MAX(4, lp / 6) PictureControlSet_t objects sized for host latencies) and -lp, md5-gated against ONE reference run.
usage: pool_sweep.py [frames] [lp,lp,...] [pool,pool,...] [mode,mode,...] [K=V+K=V,...]     modes: 1 (every covered picture) | pb (P / B pictures only) | front (no SVT_HOOK_MD);
the last argument: extra environment sets to cross with the rest (e.g. SVT_AMD_MD_MAX_KERNELS=4,SVT_HOOK_PIN_HOST=0+SVT_AMD_MD_MAX_KERNELS=64; "-" = none)"""
import json
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import encoder_fps as E
S = E.S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 320
lps = (sys.argv[2] if len(sys.argv) > 2 else "32").split(",")
pools = (sys.argv[3] if len(sys.argv) > 3 else "0,8,12,16").split(",")
modes = (sys.argv[4] if len(sys.argv) > 4 else "1").split(",")
extras = (sys.argv[5] if len(sys.argv) > 5 else "-").split(",")
w, h, depth, args = E.CONFIGS["cfg3"]
args = list(args) + ["-asm", "1"]
with tempfile.TemporaryDirectory() as td:
    yuv = os.path.join(td, "clip.yuv")
    S.write_clip(yuv, "motion", w, h, 16, 7)
    ref = {}
    for lp in lps:
        ref[lp] = E.run_app(S.REF_APP, yuv, w, h, frames, args + ["-lp", lp], os.path.join(td, "ref.265"), nb=16)
        print(json.dumps({"reference": True, "lp": int(lp), "frames": frames, "fps": ref[lp]["fps"], "md5": ref[lp]["md5"]}), flush=True)
    for lp in lps:
        for mode in modes:
            for pool, extra in ((p_, x_) for p_ in pools for x_ in extras):
                env = {} if mode == "front" else {"SVT_HOOK_MD": mode}
                if extra != "-":
                    env.update(kv.split("=", 1) for kv in extra.split("+"))
                if pool != "0":
                    env["SVT_HOOK_PCS_POOL"] = pool
                rep = os.path.join(td, "report.txt")
                env["SVT_HOOK_REPORT"] = rep
                fst = os.path.join(td, "flights.txt")
                env["SVT_AMD_MD_FLIGHT_STATS"] = fst
                try:
                    r = E.run_app(E.HIP_APP, yuv, w, h, frames, args + ["-lp", lp], os.path.join(td, "hip.265"), env=dict({"SVT_HOOK_WATCHDOG": "15"}, **env), nb=16,
                                  timeout=int(os.environ.get("SWEEP_TIMEOUT", "120")))
                except Exception as ex:  # noqa: BLE001
                    print(json.dumps({"lp": int(lp), "mode": mode, "pool": int(pool), "error": str(ex)[-2500:]}), flush=True)
                    continue
                cov = ""
                if os.path.exists(rep):
                    for line in open(rep):
                        if "mode decision:" in line and "pictures (" in line:
                            cov = line.split("mode decision:")[1].strip()[:120]
                    os.unlink(rep)
                flights = ""
                if os.path.exists(fst):
                    flights = open(fst).read().strip().replace("svt_amd: mode-decision launches: ", "")
                    os.unlink(fst)
                print(json.dumps({"lp": int(lp), "mode": mode, "pool": int(pool), "extra": extra, "frames": frames, "fps": r["fps"], "identical": r["md5"] == ref[lp]["md5"],
                                  "reference_fps": ref[lp]["fps"], "coverage": cov, "flights": flights}), flush=True)
