"""AC energy of the LCUs (an input of the mode-decision configuration) on the CPU: oracle/svt_oracle_sbo.c against the REFERENCE's own
ComputeNxMSatdSadLCU (oracle/_ref/libsvtref.so - the dispatch the encoder runs, SSE4.1 8x8 Hadamard sums) on the two block sizes CalculateAcEnergy
(Codec/EbSourceBasedOperationsProcess.c:302) asks for, and known answers."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S


def sigs(oracle):
    oracle.svt_oracle_sbo_ac_energy.restype, oracle.svt_oracle_sbo_ac_energy.argtypes = C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    oracle.svt_oracle_sbo_ac_energy_picture.restype, oracle.svt_oracle_sbo_ac_energy_picture.argtypes = None, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]


def oracle_picture(oracle, luma, w, h):
    sigs(oracle)
    luma = np.ascontiguousarray(luma)
    out = np.zeros((S.lcu_count(w, h), 5), np.uint64)
    oracle.svt_oracle_sbo_ac_energy_picture(luma.ctypes.data, luma.shape[1], w, h, out.ctypes.data)
    return out


def blocks():
    rng = np.random.default_rng(11)
    yield np.zeros((64, 96), np.uint8)
    yield np.full((64, 96), 255, np.uint8)
    yield np.tile(np.array([[0, 255], [255, 0]], np.uint8), (32, 48))          # the largest single coefficient
    for trial in range(24):
        if trial % 4 == 0:
            yield np.full((64, 96), rng.integers(0, 256), np.uint8)
        elif trial % 4 == 1:
            yield (rng.integers(0, 256, (64, 96)) >> rng.integers(0, 7)).astype(np.uint8)
        else:
            yield rng.integers(0, 256, (64, 96), dtype=np.uint8)


def test_flat_blocks_have_no_ac_energy(oracle):
    sigs(oracle)
    for v in (0, 1, 3, 128, 255):
        b = np.full((64, 64), v, np.uint8)
        # each 8x8 block: (64 v + 2) >> 2 from the Hadamard sum, the DC terms (64 * 64 v) >> 2 taken off once for the whole block
        assert oracle.svt_oracle_sbo_ac_energy(b.ctypes.data, 64, 64, 64) == 64 * ((64 * v + 2) >> 2) - ((64 * 64 * v) >> 2)


def test_oracle_matches_the_reference_symbol(oracle):
    ref = S.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libsvtref.so not built")
    sigs(oracle)
    ref.ComputeNxMSatdSadLCU.restype, ref.ComputeNxMSatdSadLCU.argtypes = C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    n = 0
    for b in blocks():
        for (x, y, s) in ((0, 0, 64), (0, 0, 32), (32, 0, 32), (0, 32, 32), (32, 32, 32), (17, 5, 32), (8, 8, 8), (24, 40, 16)):
            p = b[y:, x:]
            assert oracle.svt_oracle_sbo_ac_energy(p.ctypes.data, 96, s, s) == ref.ComputeNxMSatdSadLCU(p.ctypes.data, 96, s, s), (n, x, y, s)
        n += 1


def test_picture_form_marks_incomplete_lcus(oracle):
    w, h = 416, 240
    luma = S.gen_luma("objects", w, h, 1, 5)
    out = oracle_picture(oracle, luma, w, h)
    wl = (w + 63) // 64
    for k in range(out.shape[0]):
        x, y = 64 * (k % wl), 64 * (k // wl)
        if x + 64 > w or y + 64 > h:
            assert (out[k] == 100000000).all()
        else:
            assert out[k, 0] < 100000000 and out[k, 1:].sum() >= out[k, 0] - 3       # the four DC terms are rounded separately
    assert (out[:, 0] < 100000000).sum() == (w // 64) * (h // 64)
