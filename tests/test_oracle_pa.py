"""Picture-analysis statistics on the CPU: oracle/svt_oracle_pa.c against what the REFERENCE encoder gathered (tests/golden/pa_*.npz, recorded
by oracle/ref_harness_me_dump.c under SVT_REF_PA_DUMP) - variance / mean of the 85 blocks of every LCU, region histograms of the 1/16 picture - and
against the reference's leaf symbols on random blocks."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "pa_*.npz")))


def sigs(oracle):
    oracle.svt_oracle_pa_block_stats.restype, oracle.svt_oracle_pa_block_stats.argtypes = None, [C.c_void_p, C.c_uint32, C.c_void_p]
    oracle.svt_oracle_pa_luma_histogram.restype = C.c_uint64
    oracle.svt_oracle_pa_luma_histogram.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]


def padded_luma(kind, w, h, t, seed, pad=64):
    """the encoder's padded input picture: edges replicated (GeneratePadding); partial LCUs read it"""
    return np.ascontiguousarray(np.pad(S.gen_luma(kind, w, h, t, seed), ((0, pad), (0, pad)), mode="edge"))


def oracle_picture(oracle, luma_padded, w, h):
    sigs(oracle)
    wl, hl = (w + 63) // 64, (h + 63) // 64
    out = np.zeros(wl * hl, S.PA_LCU_STATS_DTYPE)
    for k in range(wl * hl):
        x, y = 64 * (k % wl), 64 * (k // wl)
        oracle.svt_oracle_pa_block_stats(luma_padded[y:, x:].ctypes.data, luma_padded.shape[1], out[k:k + 1].ctypes.data)
    six = np.ascontiguousarray(luma_padded[:h:4, :w:4])      # the 1/16 picture: point decimation (Decimation2D, step 4)
    hist, ravg = np.zeros((4, 4, 256), np.uint32), np.zeros((4, 4), np.uint8)
    total = oracle.svt_oracle_pa_luma_histogram(six.ctypes.data, six.shape[1], w // 4, h // 4, 4, 4, hist.ctypes.data, ravg.ctypes.data)
    return out, hist, ravg, total


def test_have_cases():
    assert len(CASES) >= 3


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_what_the_encoder_gathered(oracle, name):
    g = np.load(os.path.join(S.GOLDEN_DIR, "pa_%s.npz" % name))
    kind, w, h, n, seed = g["clip"][0], int(g["clip"][1]), int(g["clip"][2]), int(g["clip"][3]), int(g["clip"][4])
    for i, p in enumerate(g["picture_number"].tolist()):
        out, hist, ravg, total = oracle_picture(oracle, padded_luma(kind, w, h, int(p), seed), w, h)
        assert np.array_equal(out["variance"], g["variance"][i]), (name, p, np.argwhere(out["variance"] != g["variance"][i])[:4].tolist())
        assert np.array_equal(out["y_mean"], g["y_mean"][i]), (name, p)
        assert np.array_equal(hist, g["histogram"][i]), (name, p, "histogram")
        assert np.array_equal(ravg, g["region_average"][i]), (name, p)
        assert int(g["average_intensity"][i]) == (total + ((w * h) >> 1)) // (w * h), (name, p)     # CalculateInputAverageIntensity (:3960), scene-change mode 1
    assert g["variance"].max() > 100


def test_block_sums_match_the_reference_symbols(oracle):
    ref = S.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libsvtref.so not built")
    sigs(oracle)
    ref.ComputeSubMean8x8_SSE2_INTRIN.restype, ref.ComputeSubMean8x8_SSE2_INTRIN.argtypes = C.c_uint64, [C.c_void_p, C.c_uint16]
    ref.ComputeSubdMeanOfSquaredValues8x8_SSE2_INTRIN.restype, ref.ComputeSubdMeanOfSquaredValues8x8_SSE2_INTRIN.argtypes = C.c_uint64, [C.c_void_p, C.c_uint16]
    rng = np.random.default_rng(3)
    for trial in range(20):
        lcu = rng.integers(0, 256, (64, 80), dtype=np.uint8) if trial % 3 else np.full((64, 80), rng.integers(0, 256), np.uint8)
        out = np.zeros(1, S.PA_LCU_STATS_DTYPE)
        oracle.svt_oracle_pa_block_stats(lcu.ctypes.data, 80, out.ctypes.data)
        for b in range(64):
            p = lcu[8 * (b >> 3):, 8 * (b & 7):]
            m = ref.ComputeSubMean8x8_SSE2_INTRIN(p.ctypes.data, 80)
            s = ref.ComputeSubdMeanOfSquaredValues8x8_SSE2_INTRIN(p.ctypes.data, 80)
            assert int(out[0]["y_mean"][21 + b]) == (m >> 8) & 0xFF and int(out[0]["variance"][21 + b]) == ((s - m * m) >> 16) & 0xFFFF, (trial, b)
