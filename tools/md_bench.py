#!/usr/bin/env python3
"""GPU box: device-resident mode decision + encode pass of one I picture at a BASELINE size.  The unmodified reference (oracle/_ref, prebuilt)
encodes one picture with the recording harness on (SVT_REF_MD_DUMP); the device call runs on the recorded inputs, its decisions are compared
with the reference's, and the call is timed.  usage: md_bench.py [w h encMode reps]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import svtlib as S  # noqa: E402
from make_md_golden import parse_dump  # noqa: E402
from test_oracle_md_golden import compare_md  # noqa: E402


def record(w, h, enc_mode, kind="motion", extra=()):
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "c.yuv"), os.path.join(td, "md.dump")
        S.write_clip(yuv, kind, w, h, 1, 7)
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", "1", "-asm", "1", "-b", os.path.join(td, "o.265"), "-encMode", str(enc_mode),
               "-intra-period", "0", "-q", "32"] + list(extra)
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_MD_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        pics, lcus = parse_dump(open(dump, "rb").read())
    h0, y, cb, cr, ois = pics[0]
    r = lcus[lcus["picture_number"] == 0]
    r = r[np.argsort(r["lcu_index"])]
    return dict(pic=np.array([h0["pic"]]), cost=np.ascontiguousarray(h0["cost"]), y=np.ascontiguousarray(y), cb=np.ascontiguousarray(cb),
                cr=np.ascontiguousarray(cr), ois=np.ascontiguousarray(ois), lcu=np.ascontiguousarray(r["lcu"]), out=np.ascontiguousarray(r["out"]))


def run(lib, g, reps=5, check=True):
    from test_gpu_md import sig
    sig(lib)
    w, h = int(g["pic"][0]["width"]), int(g["pic"][0]["height"])
    ctx, pic = C.c_void_p(), C.c_void_p()
    assert lib.svt_amd_context_create(0, w, (h + 7) & ~7, 2, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    assert lib.svt_amd_encdec_picture_create(ctx, w, h, 1, C.byref(pic)) == 0, lib.svt_amd_last_error()
    n = len(g["lcu"])
    out, works, res = np.zeros(n, S.MD_LCU_OUT_DTYPE), np.zeros(n, S.LCU_WORK_DTYPE), np.zeros(n, S.LCU_RESULT_DTYPE)
    ts = []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rc = lib.svt_amd_md_encode_picture(ctx, pic, g["pic"].ctypes.data, g["lcu"].ctypes.data, g["y"].ctypes.data, w, g["cb"].ctypes.data, g["cr"].ctypes.data,
                                           w // 2, g["ois"].ctypes.data, 0, g["cost"].ctypes.data, out.ctypes.data, works.ctypes.data, res.ctypes.data)
        assert rc == 0, lib.svt_amd_last_error()
        ts.append((time.perf_counter() - t0) * 1e3)
    if check:
        compare_md(out, g["out"], "%dx%d" % (w, h))
    lib.svt_amd_encdec_picture_destroy(ctx, pic)
    lib.svt_amd_context_destroy(ctx)
    units = int(works["num_cus"].sum())
    tested = int(g["out"]["tested"].sum())
    return {"width": w, "height": h, "lcus": n, "leaves_tested": tested, "final_units": units, "ms_first_call": round(ts[0], 2),
            "ms_per_picture": round(float(np.median(ts[1:])), 2), "pictures_per_s_one_in_flight": round(1e3 / float(np.median(ts[1:])), 2),
            "what": "svt_amd_md_encode_picture through the host-array ABI (source planes, OIS and LCU controls up, decisions + work + result records down), "
                    "decisions identical to the reference's ModeDecisionLcu records of the same picture"}


if __name__ == "__main__":
    a = sys.argv[1:]
    w, h, m, reps = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (3840, 2160, 7, 5)
    g = record(w, h, m)
    print(json.dumps(run(S.load_product(), g, reps)))
