"""CPU-only: pins oracle/svt_oracle_intra.c:svt_oracle_intra_pu on the intra 4x4 coding units of the encode pass
(EbCodingLoop.c:3594-3690: GenerateLumaIntraReferenceSamplesEncodePass per 4x4 partition, GenerateChromaIntraReferenceSamplesEncodePass
once per 8x8 coding unit, EncodePassIntraPrediction with the luma / chroma mask; tests/golden/intra4_*.npz, made by
tests/golden/make_intra_golden.py i4:<name>): a size-4 job predicts the luma partition, a size-8 job the 4x4 chroma pair."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S
from test_oracle_intra_golden import job_of, want_of

CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "intra4_*.npz")))


def load_intra4_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "intra4_%s.npz" % name)))
    n = g["size"].astype(np.int64)
    g["off_y"] = np.concatenate([[0], np.cumsum(n ** 2)])
    g["off_c"] = np.concatenate([[0], np.cumsum((n // 2) ** 2)])
    return g


def test_have_cases():
    assert len(CASES) == 3


@pytest.mark.parametrize("name", CASES)
def test_intra_pu_oracle_matches_intra4x4_records(oracle, name):
    g = load_intra4_case(name)
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    oracle.svt_oracle_intra_pu.restype = None
    seen = set()
    for i in range(len(g["size"])):
        bps, size, mask = int(g["bytes_per_sample"][i]), int(g["size"][i]), int(g["component_mask"][i])
        assert (size, mask) in ((4, 1), (8, 6))
        want = want_of(g, i)
        got = [np.zeros_like(w) for w in want]
        j = job_of(g, i)
        oracle.svt_oracle_intra_pu(bps, j.ctypes.data, got[0].ctypes.data, got[0].shape[1], got[1].ctypes.data, got[2].ctypes.data,
                                   got[1].shape[1])
        for p in ([0] if mask == 1 else [1, 2]):
            assert np.array_equal(got[p], want[p]), (name, i, p, size, int(g["luma_mode"][i]), got[p].tolist(), want[p].tolist())
        seen.add((mask, int(g["luma_mode"][i])))
    assert len([m for m in seen if m[0] == 1]) >= 20 and len([m for m in seen if m[0] == 6]) >= 5
