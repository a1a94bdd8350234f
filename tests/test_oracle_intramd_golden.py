"""CPU-only: pins oracle/svt_oracle_intra.c:svt_oracle_intra_pu against records of the mode decision's closed-loop intra
prediction (IntraPredictionCl with GenerateIntraLuma/ChromaReferenceSamplesMd, Codec/EbIntraPrediction.c:3682,
EbProductCodingLoop.c:269, :2196; tests/golden/intramd_*.npz, made by tests/golden/make_intra_golden.py md:<name>): one record
per luma block (component_mask 1) or chroma pair (component_mask 6), cut from the mode decision's own neighbour arrays."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S
from test_oracle_intra_golden import job_of, want_of

CASES = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "intramd_*.npz")))
# the open-loop twin IntraPredictionOl (:5427): neighbours = source samples (UpdateNeighborSamplesArrayOL), no smoothing
OL_CASES = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "intraol_*.npz")))


def load_intramd_case(name, ol=False):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, ("intraol_%s.npz" if ol else "intramd_%s.npz") % name)))
    n = g["size"].astype(np.int64)
    g["off_y"] = np.concatenate([[0], np.cumsum(n ** 2)])
    g["off_c"] = np.concatenate([[0], np.cumsum((n // 2) ** 2)])
    return g


def planes_of(mask):
    return [0] if mask == 1 else [1, 2]


def md_job_of(g, i):
    j = job_of(g, i)
    if "no_smoothing" in g:
        j["no_smoothing"] = g["no_smoothing"][i]
    return j


def test_have_cases():
    assert len(CASES) == 4 and len(OL_CASES) == 3


@pytest.mark.parametrize("name,ol", [(n, False) for n in CASES] + [(n, True) for n in OL_CASES])
def test_intra_pu_oracle_matches_mode_decision_records(oracle, name, ol):
    g = load_intramd_case(name, ol)
    assert (not ol) or (g["no_smoothing"] == 1).all()
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    oracle.svt_oracle_intra_pu.restype = None
    seen = set()
    for i in range(len(g["size"])):
        want = want_of(g, i)
        got = [np.zeros_like(w) for w in want]
        j = md_job_of(g, i)
        oracle.svt_oracle_intra_pu(1, j.ctypes.data, got[0].ctypes.data, got[0].shape[1], got[1].ctypes.data, got[2].ctypes.data,
                                   got[1].shape[1])
        for p in planes_of(int(g["component_mask"][i])):
            assert np.array_equal(got[p], want[p]), (name, i, p, int(g["size"][i]), int(g["luma_mode"][i]),
                                                     np.argwhere(got[p] != want[p])[:4].tolist())
        seen.add((int(g["component_mask"][i]), int(g["size"][i])))
    assert ({(1, 16), (1, 32)} if ol else {(1, 8), (1, 16), (1, 32)}).issubset(seen)
