"""CPU-only: pins oracle/svt_oracle_leaf.c:svt_oracle_fast_loop_distortion (luma SAD, Cb + Cr SAD with chroma in the loop,
nothing for most-probable-mode candidates) against records of real second-loop iterations of ProductPerformFastLoop
(tests/golden/fastloop_*.npz, made by tests/golden/make_fastloop_golden.py): the source and predicted blocks in, the two
distortions the loop handed to the fast-cost function out.  Records of the two cases the caller keeps (the candidate that reuses
its open-loop distortion, the quartered chroma of noise LCUs) are recognised and checked against their own rule."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[9:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "fastloop_*.npz")))
CAND = np.dtype([("src_off_y", "<i4"), ("src_off_c", "<i4"), ("pred_off_y", "<i4"), ("pred_off_c", "<i4"), ("size", "u1"), ("flags", "u1"),
                 ("pad", "u1", 2)])
DIST = np.dtype([("luma", "<u4"), ("chroma", "<u4")])


def load_fastloop_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "fastloop_%s.npz" % name)))
    n = g["size"].astype(np.int64)
    g["off_y"] = np.concatenate([[0], np.cumsum(n ** 2)])
    g["off_c"] = np.concatenate([[0], np.cumsum((n // 2) ** 2)])
    return g


def blocks_of(g, i):
    n = int(g["size"][i])
    a, b, c, d = int(g["off_y"][i]), int(g["off_y"][i + 1]), int(g["off_c"][i]), int(g["off_c"][i + 1])
    return ([np.ascontiguousarray(g["src_" + p][(a if p == "y" else c):(b if p == "y" else d)]).reshape(-1, n if p == "y" else n // 2)
             for p in ("y", "cb", "cr")],
            [np.ascontiguousarray(g["pred_" + p][(a if p == "y" else c):(b if p == "y" else d)]).reshape(-1, n if p == "y" else n // 2)
             for p in ("y", "cb", "cr")])


def cand_of(g, i):
    k = np.zeros(1, CAND)
    k["size"] = g["size"][i]
    k["flags"] = (1 if g["use_chroma"][i] else 0) | (2 if g["mpm_flag"][i] else 0)
    return k


def reference_rule(g, i, dist):
    """what ProductPerformFastLoop hands over, given the plain measurement: (luma, chroma) or None when the record is one of the
    caller-side special cases whose condition the record cannot decide"""
    luma, chroma = int(dist["luma"]), int(dist["chroma"])
    want_l, want_c = int(g["luma_distortion"][i]), int(g["chroma_distortion"][i])
    if luma != want_l:
        # :2042-2043: the best candidate of the first loop, when intra, keeps its open-loop distortion; the recorder also lets
        # through a few first-loop calls (:1968, open-loop distortion, no prediction) that reuse a buffer predicted earlier
        assert int(g["distortion_ready"][i]) and want_l == int(g["me_distortion"][i]), i
        luma = want_l
    if chroma != want_c:
        # :2079-2090: zero-motion 64x64 candidates of noise LCUs
        assert int(g["noise_lcu"][i]) and int(g["size"][i]) == 64 and want_c == chroma >> 2, i
        chroma = want_c
    return luma, chroma


def test_layout_and_cases():
    assert CAND.itemsize == 20 and DIST.itemsize == 8 and len(CASES) == 4


@pytest.mark.parametrize("name", CASES)
def test_fast_loop_distortion_oracle_matches_reference(oracle, name):
    g = load_fastloop_case(name)
    oracle.svt_oracle_fast_loop_distortion.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                       C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    oracle.svt_oracle_fast_loop_distortion.restype = None
    plain = special = 0
    for i in range(len(g["size"])):
        src, pred = blocks_of(g, i)
        n = int(g["size"][i])
        d, k = np.zeros(1, DIST), cand_of(g, i)
        oracle.svt_oracle_fast_loop_distortion(k.ctypes.data, src[0].ctypes.data, n, src[1].ctypes.data, src[2].ctypes.data, n // 2,
                                               pred[0].ctypes.data, n, pred[1].ctypes.data, pred[2].ctypes.data, n // 2, d.ctypes.data)
        got = (int(d[0]["luma"]), int(d[0]["chroma"]))
        want = (int(g["luma_distortion"][i]), int(g["chroma_distortion"][i]))
        if got == want:
            plain += 1
        else:
            assert reference_rule(g, i, d[0]) == want, (name, i)
            special += 1
    assert plain >= 0.8 * len(g["size"]), (plain, special)
