#!/usr/bin/env python3
"""Whole-encoder fps of the hooked encoder (integration/_build/SvtHevcEncApp_hip: reference host control + the HIP
hot path behind the bindings) next to the unmodified reference (oracle/_ref/SvtHevcEncApp_ref, -asm 1 = its AVX2 path)
on the same synthetic clip, md5-gated: the BASELINE.json metric ("encoded fps ... bit-exact").

fps is the application's own "Average Speed" line (Source/App/EbAppProcessCmd.c:1419: frames / encode wall time, start-up
excluded); the clip is preloaded with -nb so file reads are outside it.

usage: encoder_fps.py [cfg2|cfg3|cfg4|WxH] [frames] [extra app args ...]      env: SVT_HOOK_* switches as the hook reads them
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402

HIP_APP = os.path.join(ROOT, "integration", "_build", "SvtHevcEncApp_hip")

CONFIGS = {
    # BASELINE.md section 2 command lines
    "cfg2": (1920, 1080, 8, ["-encMode", "9", "-pred-struct", "0", "-q", "32"]),
    "cfg3": (3840, 2160, 8, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60", "-q", "32"]),
    "cfg4": (3840, 2160, 10, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60", "-q", "32",
                              "-bit-depth", "10", "-compressed-ten-bit-format", "1"]),
}


def run_app(app, yuv, w, h, n, args, out, env=None, timeout=300, nb=None):
    """nb: frames preloaded into memory before the clock starts (-nb, Source/App/EbAppContext.c:420); the application cycles
    through them, so n may exceed the frames the file holds."""
    cmd = [app, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-nb", str(nb or n), "-b", out] + args
    t0 = time.perf_counter()
    child_env = dict(os.environ, **(env or {}))
    if not (env and "GPU_MAX_HW_QUEUES" in env):
        child_env.pop("GPU_MAX_HW_QUEUES", None)   # the encoder process takes the library's own default (bench.py sets a smaller one for its in-process loops)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=child_env)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("encoder failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout[-1500:], r.stderr[-1500:]))
    if os.environ.get("SVT_TOOLS_STDERR_TO") and app == HIP_APP:   # diagnosis: keep what the bindings wrote to stderr
        open(os.environ["SVT_TOOLS_STDERR_TO"], "w").write(r.stderr)
    fps = None
    for line in r.stdout.splitlines():
        if "Average Speed" in line:
            fps = float(line.split()[2])
    md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    return {"fps": fps, "wall_s": round(wall, 2), "md5": md5, "bytes": os.path.getsize(out)}


def measure(cfg="cfg3", frames=32, extra=(), asm="1", tmpdir=None, hip_env=None, unique=None):
    """unique: distinct frames written to the clip (default: all); the encoders cycle through them for `frames` pictures."""
    if cfg in CONFIGS:
        w, h, depth, args = CONFIGS[cfg]
    else:
        w, h = (int(v) for v in cfg.split("x"))
        depth, args = 8, []
    args = list(args) + list(extra)
    if "-asm" not in args:
        args += ["-asm", asm]
    with tempfile.TemporaryDirectory(dir=tmpdir) as td:
        yuv = os.path.join(td, "clip.yuv")
        nu = min(unique or frames, frames)
        if depth == 10 and "-compressed-ten-bit-format" in args:
            S.write_clip10_compressed(yuv, "motion", w, h, nu, 7)
        elif depth == 10:
            S.write_clip10(yuv, "motion", w, h, nu, 7)
        else:
            S.write_clip(yuv, "motion", w, h, nu, 7)
        ref = run_app(S.REF_APP, yuv, w, h, frames, args, os.path.join(td, "ref.265"), nb=nu)
        hip = run_app(HIP_APP, yuv, w, h, frames, args, os.path.join(td, "hip.265"), env=hip_env, nb=nu)
    return {"config": cfg, "width": w, "height": h, "frames": frames, "unique_frames": nu, "args": " ".join(args), "host_threads": os.cpu_count(),
            "reference": ref, "hip": hip, "bitstream_identical": ref["md5"] == hip["md5"],
            "hip_over_reference": round(hip["fps"] / ref["fps"], 3) if ref["fps"] and hip["fps"] else None}


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    print(json.dumps(measure(cfg, frames, sys.argv[3:])))
