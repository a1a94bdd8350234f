"""CPU-only: pins oracle/svt_oracle_intra.c:svt_oracle_intra_pu (neighbour availability + substitution + smoothing + mode
dispatch + prediction of the three blocks) against records of real GenerateIntraReferenceSamplesEncodePass +
EncodePassIntraPrediction call pairs of the reference's encode pass (tests/golden/intra_*.npz, made by
tests/golden/make_intra_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "intra_*.npz")))

JOB = np.dtype([("size", "<u4"), ("constrained_intra", "u1"), ("strong_smoothing", "u1"), ("pic_left", "u1"), ("pic_top", "u1"),
                ("pic_right", "u1"), ("bottom_left_ok", "u1"), ("top_right_ok", "u1"), ("luma_mode", "u1"), ("chroma_mode", "u1"),
                ("mode_tl", "u1"), ("mode_left", "u1", 16), ("mode_top", "u1", 16), ("no_smoothing", "u1"), ("pad", "u1"), ("left", "<u2", (3, 64)),
                ("top", "<u2", (3, 64)), ("tl", "<u2", 3), ("pad2", "<u2"), ("dst_off_y", "<i4"), ("dst_off_c", "<i4")])


def load_intra_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "intra_%s.npz" % name)))
    n = g["size"].astype(np.int64)
    g["off_y"] = np.concatenate([[0], np.cumsum(n ** 2)])
    g["off_c"] = np.concatenate([[0], np.cumsum((n // 2) ** 2)])
    return g


def job_of(g, i):
    j = np.zeros(1, JOB)
    for k in ("size", "constrained_intra", "strong_smoothing", "pic_left", "pic_top", "pic_right", "bottom_left_ok", "top_right_ok",
              "luma_mode", "chroma_mode", "mode_tl", "tl"):
        j[k] = g[k][i]
    j["mode_left"], j["mode_top"] = g["mode_left"][i][:16], g["mode_top"][i][:16]
    j["left"], j["top"] = g["left"][i][:, :64], g["top"][i][:, :64]
    return j


def want_of(g, i):
    n, bps = int(g["size"][i]), int(g["bytes_per_sample"][i])
    dt = np.uint8 if bps == 1 else np.uint16
    a, b = int(g["off_y"][i]), int(g["off_y"][i + 1])
    c, d = int(g["off_c"][i]), int(g["off_c"][i + 1])
    return (g["pred_y"][a:b].reshape(n, n).astype(dt), g["pred_cb"][c:d].reshape(n // 2, n // 2).astype(dt),
            g["pred_cr"][c:d].reshape(n // 2, n // 2).astype(dt))


def test_struct_size():
    assert JOB.itemsize == 4 + 10 + 32 + 2 + 768 + 8 + 8 == 832


def test_have_cases():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_intra_pu_oracle_matches_reference(oracle, name):
    g = load_intra_case(name)
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    oracle.svt_oracle_intra_pu.restype = None
    modes = set()
    for i in range(len(g["size"])):
        if int(g["size"][i]) > 32:
            continue
        bps = int(g["bytes_per_sample"][i])
        want = want_of(g, i)
        got = [np.zeros_like(w) for w in want]
        j = job_of(g, i)
        oracle.svt_oracle_intra_pu(bps, j.ctypes.data, got[0].ctypes.data, got[0].shape[1], got[1].ctypes.data, got[2].ctypes.data,
                                   got[1].shape[1])
        for p in range(3):
            assert np.array_equal(got[p], want[p]), (name, i, p, int(g["size"][i]), int(g["luma_mode"][i]),
                                                     np.argwhere(got[p] != want[p])[:4].tolist())
        modes.add(int(g["luma_mode"][i]))
    assert len(modes) >= 10
