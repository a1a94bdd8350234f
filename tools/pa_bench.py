#!/usr/bin/env python3
"""Picture-analysis statistics at 4K (needs the GPU): svt_amd_picture_stats on a picture resident in a slot; ms per call (blocking, incl. the 0.5 MB of records
coming back).  Kernel times: run under rocprofv3 --kernel-trace --stats.  usage: python tools/pa_bench.py [iters]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    w, h = 3840, 2160
    lib = S.load_product()
    lib.svt_amd_picture_stats.restype = C.c_int
    lib.svt_amd_picture_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    luma = np.ascontiguousarray(S.gen_luma("motion", w, h, 1, 7))
    assert lib.svt_amd_picture_upload(ctx, 0, luma.ctypes.data, w, w, h) == 0
    out = np.zeros(S.lcu_count(w, h), S.PA_LCU_STATS_DTYPE)
    hist, ravg, total = np.zeros((4, 4, 256), np.uint32), np.zeros((4, 4), np.uint8), C.c_uint64(0)
    args = (ctx, 0, out.ctypes.data, 4, 4, hist.ctypes.data, ravg.ctypes.data, C.byref(total))
    for _ in range(3):
        assert lib.svt_amd_picture_stats(*args) == 0, lib.svt_amd_last_error()
    t0 = time.perf_counter()
    for _ in range(iters):
        lib.svt_amd_picture_stats(*args)
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(json.dumps({"picture": "%dx%d" % (w, h), "ms_per_call": round(ms, 4), "lcus": len(out), "bytes_read_algorithmic": w * h // 2 + w * h // 16,
                      "what": "block means / variances of every LCU (even rows of the luma plane) + 4 x 4 region histograms of the 1/16 plane; blocking call incl. records to the host"}))
    lib.svt_amd_context_destroy(ctx)


if __name__ == "__main__":
    main()
