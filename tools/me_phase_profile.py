#!/usr/bin/env python3
"""Per-phase time breakdown of k_me_picture (shader-clock stamps taken by thread 0 of each
workgroup).  usage: python tools/me_phase_profile.py [batch] [fixture, e.g. b_3840x2160_m7]   (needs the GPU)"""
import re
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402
from golden_util import load_case  # noqa: E402

# stamp i .. i+1 (k_me<0> writes 0..4, k_me<1> writes 5..12; 4 -> 5 is the gap between the two kernels)
NAMES = ["hme: stage src", "hme: TestSearchAreaBounds", "hme: HME L0/L1/L2", "hme: CheckZeroZero", None,
         "search: stage + init", "search: full-pel (+windows)", "search: SuPelEnable", "search: half-pel",
         "search: quarter-pel", "search: bi-pred", "search: records"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    case = sys.argv[2] if len(sys.argv) > 2 else "p_1920x1080_m9"
    W, H = (int(v) for v in re.search(r"_(\d+)x(\d+)_", case).groups())
    lib = S.load_product()
    lib.svt_amd_debug_me_phase_profile.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, W, (H + 7) & ~7, B + 2, C.byref(ctx)) == 0
    g = load_case(case)
    params = S.params_from_record(g["params"][len(g["params"]) // 2])
    print("%s: %dx%d, %d list(s), batch %d" % (case, W, H, params.num_lists, B))
    dev = torch.device("cuda", 0)
    frames = torch.randint(0, 256, (B + 2, H, W), dtype=torch.uint8, device=dev)
    for t in range(1, B + 2):  # smooth-ish motion: shifted copies + noise
        frames[t] = torch.roll(frames[0], shifts=(t, 2 * t), dims=(0, 1))
    torch.cuda.synchronize()
    for i in range(B + 2):
        assert lib.svt_amd_picture_upload_device(ctx, i, C.c_void_p(frames[i].data_ptr()), W, W, H) == 0
    jobs = (S.MeJob * B)()
    for i in range(B):
        jobs[i].params, jobs[i].cur_slot = params, i + 1
        jobs[i].ref_slot[0], jobs[i].ref_slot[1] = i, i + 2
    nlcu = S.lcu_count(W, H)
    for _ in range(2):
        assert lib.svt_amd_me_batch_launch(ctx, jobs, B) == 0
    lib.svt_amd_synchronize(ctx)
    n = B * nlcu
    assert lib.svt_amd_debug_me_phase_profile(ctx, n, None) == 0
    assert lib.svt_amd_me_batch_launch(ctx, jobs, B) == 0
    buf = np.zeros((n, 16), np.uint64)
    assert lib.svt_amd_debug_me_phase_profile(ctx, n, buf.ctypes.data) == 0
    d = np.diff(buf[:, :13].astype(np.int64), axis=1)
    for k0, k1, nm in ((0, 4, "k_me<0> (hme)"), (5, 12, "k_me<1> (search)")):
        tot = (buf[:, k1] - buf[:, k0]).astype(np.int64)
        print("%s: workgroups %d; clocks per workgroup: median %d, mean %d" % (nm, n, np.median(tot), tot.mean()))
        for i in range(k0, k1):
            if NAMES[i]:
                print("  %-30s median %8d  mean %8d  (%4.1f%%)" % (NAMES[i], np.median(d[:, i]), d[:, i].mean(),
                                                                   100.0 * d[:, i].mean() / tot.mean()))
        if k0 == 5 and buf[:, 13].any():  # finer marks inside the full-pel stage
            b = buf.astype(np.int64)
            print("      stamps 13 / 14 present in %d / %d of %d workgroups; first rows %s" % ((b[:, 13] != 0).sum(), (b[:, 14] != 0).sum(), n, np.nonzero(b[:, 13])[0][:12]))
            ok = (b[:, 13] != 0) & (b[:, 14] != 0)
            b = b[ok]
            marks = ((6, 13, "F window in LDS"), (13, 14, "search proper + minima"), (14, 7, "wait for the b / h / j windows"))
            if os.environ.get("ME_EXP_STAMPS"):  # an experimental build (-DME_EXP >= 10) reuses the HME kernel's slots for finer marks
                marks = ((6, 13, "F window in LDS"), (13, 15, "issue the b / h / j LDS-DMA"), (15, 0, "search loop"), (0, 1, "64x64 pass"),
                         (1, 2, "minima across lanes"), (2, 14, "winners to LDS"), (14, 7, "wait for the b / h / j windows"))
            for a0, a1, nm2 in marks:
                v = b[:, a1] - b[:, a0]
                print("      %-32s median %8d  mean %8d" % (nm2, np.median(v), v.mean()))
        span = int(buf[:, k1].max() - buf[:, k0].min())
        print("  kernel span %d clocks; mean concurrent workgroups %.0f" % (span, tot.sum() / span))
    lib.svt_amd_context_destroy(ctx)


if __name__ == "__main__":
    main()
