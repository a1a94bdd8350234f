"""CPU-only: pins the picture-level deblocking of oracle/svt_oracle_loopfilter.c (svt_oracle_dlf_picture) against whole
pictures of real encoder runs: reconstruction before the filter + boundary strengths + qp array -> reconstruction after
the reference's three per-LCU drivers have run over every LCU (tests/golden/dlf_*.npz, made by
tests/golden/make_dlf_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "dlf_*.npz")))


def load_dlf_case(name):
    """-> list of pictures: dict(hdr, pre=[y,cb,cr], post=[y,cb,cr], bsv, bsh, qp)"""
    g = np.load(os.path.join(S.GOLDEN_DIR, "dlf_%s.npz" % name))
    pics = []
    for k in range(int(g["count"])):
        pics.append(dict(hdr=g["hdr%d" % k][0], pre=[g["pre_%s%d" % (c, k)] for c in ("y", "cb", "cr")],
                         post=[g["post_%s%d" % (c, k)] for c in ("y", "cb", "cr")], bsv=g["bsv%d" % k], bsh=g["bsh%d" % k],
                         qp=g["qp%d" % k], final=[g["final_%s%d" % (c, k)] for c in ("y", "cb", "cr")],
                         sao_flag=g["sao_flag%d" % k], sao_lcu=g["sao_lcu%d" % k], cumap=g["cumap%d" % k], cbf=g["cbf%d" % k],
                         refpoc=g["refpoc%d" % k], lcu_edge=g["lcu_edge%d" % k]))
    return pics


def oracle_dlf(oracle, pic):
    h = pic["hdr"]
    planes = [np.ascontiguousarray(p).copy() for p in pic["pre"]]
    oracle.svt_oracle_dlf_picture.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                              C.c_int, C.c_int]
    oracle.svt_oracle_dlf_picture.restype = None
    bsv, bsh, qp = (np.ascontiguousarray(pic[k]) for k in ("bsv", "bsh", "qp"))
    oracle.svt_oracle_dlf_picture(int(h["bytes_per_sample"]), planes[0].ctypes.data, planes[0].shape[1], planes[1].ctypes.data,
                                  planes[2].ctypes.data, planes[1].shape[1], int(h["width"]), int(h["height"]),
                                  bsv.ctypes.data, bsh.ctypes.data, qp.ctypes.data, int(h["qp_stride"]), int(h["tc_offset"]),
                                  int(h["beta_offset"]), int(h["cb_qp_offset"]), int(h["cr_qp_offset"]))
    return planes


def test_have_cases():
    assert len(CASES) >= 5


@pytest.mark.parametrize("name", CASES)
def test_dlf_picture_oracle_matches_reference(oracle, name):
    pics = load_dlf_case(name)
    assert len(pics) >= 2
    for k, pic in enumerate(pics):
        got = oracle_dlf(oracle, pic)
        for p in range(3):
            bad = np.argwhere(got[p] != pic["post"][p])
            assert len(bad) == 0, (name, k, p, len(bad), bad[:5].tolist())
        assert sum(int((a != b).sum()) for a, b in zip(pic["pre"], pic["post"])) > 1000   # the filter did something


def oracle_sao(oracle, src, bps, width, height, lcus, luma_on, chroma_on):
    """src: [y, cb, cr] deblocked planes -> planes after SAO"""
    src = [np.ascontiguousarray(p) for p in src]
    dst = [np.zeros_like(p) for p in src]
    lcus = np.ascontiguousarray(lcus)
    oracle.svt_oracle_sao_apply_picture.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_void_p, C.c_int, C.c_int]
    oracle.svt_oracle_sao_apply_picture.restype = None
    ps, pd = (C.c_void_p * 3)(*[a.ctypes.data for a in src]), (C.c_void_p * 3)(*[a.ctypes.data for a in dst])
    oracle.svt_oracle_sao_apply_picture(bps, ps, pd, src[0].shape[1], src[1].shape[1], width, height, lcus.ctypes.data,
                                        int(luma_on), int(chroma_on))
    return dst


@pytest.mark.parametrize("name", CASES)
def test_sao_apply_picture_oracle_matches_reference(oracle, name):
    """deblocked picture + every LCU's final SAO parameters -> the encoder's own reconstruction output"""
    seen = set()
    for k, pic in enumerate(load_dlf_case(name)):
        h = pic["hdr"]
        got = oracle_sao(oracle, pic["post"], int(h["bytes_per_sample"]), int(h["width"]), int(h["height"]), pic["sao_lcu"],
                         pic["sao_flag"][0], pic["sao_flag"][1])
        for p in range(3):
            bad = np.argwhere(got[p] != pic["final"][p])
            assert len(bad) == 0, (name, k, p, len(bad), bad[:5].tolist())
        seen |= set(np.unique(pic["sao_lcu"]["type"]).tolist())
    if name in ("p_416x240_m9", "tiles_640x384_m9"):
        assert seen >= {1, 2, 3, 4}, seen


def oracle_bs(oracle, pic):
    h = pic["hdr"]
    oracle.svt_oracle_bs_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
    oracle.svt_oracle_bs_picture.restype = None
    cumap, cbf, poc, edge = (np.ascontiguousarray(pic[k]) for k in ("cumap", "cbf", "refpoc", "lcu_edge"))
    bsv, bsh = np.zeros_like(pic["bsv"]), np.zeros_like(pic["bsh"])
    oracle.svt_oracle_bs_picture(cumap.ctypes.data, cbf.ctypes.data, int(h["width"]), int(h["height"]), int(h["slice_type"]),
                                 poc.ctypes.data, edge.ctypes.data, bsv.ctypes.data, bsh.ctypes.data)
    return bsv, bsh


@pytest.mark.parametrize("name", CASES)
def test_bs_picture_oracle_matches_reference(oracle, name):
    """coding-unit map + cbf map -> the boundary-strength arrays the reference's per-CU derivation left"""
    for k, pic in enumerate(load_dlf_case(name)):
        bsv, bsh = oracle_bs(oracle, pic)
        for nm, got, want in (("vertical", bsv, pic["bsv"]), ("horizontal", bsh, pic["bsh"])):
            bad = np.argwhere(got != want)
            assert len(bad) == 0, (name, k, nm, len(bad), [(int(a), int(b), int(got[a, b]), int(want[a, b])) for a, b in bad[:8]])
