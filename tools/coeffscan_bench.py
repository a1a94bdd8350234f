#!/usr/bin/env python3
"""Entropy hand-off pre-scan at BASELINE configs[2]'s picture size (needs the GPU): svt_amd_coeff_scan_picture on a seeded 4K picture whose encode-pass
records are resident in HBM.  Prints one JSON line: ms per call (host clock, blocking call incl. the copies of the produced lists), what was produced
against what the host would otherwise read.  Kernel times: run under rocprofv3 --kernel-trace --stats (profiles/).
usage: python tools/coeffscan_bench.py [width height] [iters]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402
from test_gpu_coeffscan import device_scan, random_records, sig  # noqa: E402


def main():
    import torch
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lib = S.load_product()
    sig(lib)
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, w, h, 1, C.byref(ctx)) == 0, lib.svt_amd_last_error()
    out = {"picture": "%dx%d" % (w, h)}
    for density, tag in ((0.15, "sparse"), (0.6, "dense")):
        works, results = random_records(w, h, 31, density)
        n = len(works)
        dw, dr = torch.from_numpy(works.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(results.view(np.uint8).reshape(-1)).cuda()
        torch.cuda.synchronize()
        args = (lib, ctx, dw.data_ptr(), works.dtype.itemsize, dr.data_ptr(), results.dtype.itemsize, 1, n)
        rc, lcus, groups, levels, totals = device_scan(*args)
        assert rc == 0, lib.svt_amd_last_error()
        t0 = time.perf_counter()
        for _ in range(iters):
            device_scan(*args)
        ms = (time.perf_counter() - t0) / iters * 1e3
        produced = n * S.COEFF_SCAN_LCU_DTYPE.itemsize + totals[0] * 8 + totals[1] * 2
        out[tag] = {"ms_per_call": round(ms, 3), "lcus": n, "groups": totals[0], "levels": totals[1], "bytes_produced": produced,
                    "bytes_of_s16_planes": w * h * 3, "ratio": round(produced / (w * h * 3), 3)}
    lib.svt_amd_context_destroy(ctx)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
