#!/usr/bin/env python3
"""GPU box: the hooked encoder with SVT_HOOK_MD (+ _VERIFY) on a small clip; prints the hook's report and whether the bitstream matches.
usage: md_verify.py kind w h n [verify|serve] [encoder args ...]"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S
HIP_APP = os.path.join(ROOT, "integration", "_build", "SvtHevcEncApp_hip")
kind, w, h, n, how = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
args = sys.argv[6:]
with tempfile.TemporaryDirectory() as td:
    yuv = os.path.join(td, "c.yuv")
    S.write_clip(yuv, kind, w, h, n, 7)
    def run(app, env, out):
        r = subprocess.run([app, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-b", out, "-asm", "1"] + args, capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=900)
        return hashlib.md5(open(out, "rb").read()).hexdigest() if os.path.exists(out) else None, r
    ref, _ = run(S.REF_APP, {}, os.path.join(td, "ref.265"))
    env = {"SVT_HOOK_MD": "1", "SVT_HOOK_REPORT": os.path.join(td, "rep.txt")}
    if how == "verify":
        env["SVT_HOOK_MD_VERIFY"] = "1"
    hip, r = run(HIP_APP, env, os.path.join(td, "hip.265"))
    print("identical" if ref == hip else "DIFFERENT", ref, hip, "rc", r.returncode)
    print("\n".join(l for l in r.stderr.splitlines() if "VERIFY" in l or l.startswith("    leaf"))[:6000])
    if os.path.exists(env["SVT_HOOK_REPORT"]):
        print(open(env["SVT_HOOK_REPORT"]).read().split("\n")[0])
