"""CPU-only: pins oracle/svt_oracle_fullloop.c:svt_oracle_recon_tu (inverse transform or DC shortcut + prediction, clipped)
against records of real EncodeGenerateRecon(16bit) calls of the reference's encode pass (tests/golden/recon_*.npz, made by
tests/golden/make_recon_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "recon_*.npz")))


def load_recon_case(name):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, "recon_%s.npz" % name)))
    g["offsets"] = np.concatenate([[0], np.cumsum(g["size"].astype(np.int64) ** 2)])
    return g


def record(g, i):
    a, b, n = int(g["offsets"][i]), int(g["offsets"][i + 1]), int(g["size"][i])
    bps = int(g["bytes_per_sample"][i])
    dt = np.uint8 if bps == 1 else np.uint16
    return (n, bps, int(g["only_dc"][i]), int(g["dst"][i]), np.ascontiguousarray(g["coeff"][a:b]).reshape(n, n),
            g["pred"][a:b].reshape(n, n).astype(dt), g["recon"][a:b].reshape(n, n).astype(dt))


def test_have_cases():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_recon_oracle_matches_reference(oracle, name):
    g = load_recon_case(name)
    oracle.svt_oracle_recon_tu.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_uint32]
    oracle.svt_oracle_recon_tu.restype = None
    seen = set()
    for i in range(len(g["size"])):
        n, bps, only_dc, dst, coeff, pred, want = record(g, i)
        got = np.zeros_like(pred)
        oracle.svt_oracle_recon_tu(bps, n, only_dc, dst, coeff.ctypes.data, pred.ctypes.data, n, got.ctypes.data, n)
        assert np.array_equal(got, want), (name, i, n, bps, only_dc, dst)
        seen.add((n, only_dc, dst))
    assert len(seen) >= 4
