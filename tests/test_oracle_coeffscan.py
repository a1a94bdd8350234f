"""Entropy hand-off pre-scan on the CPU: the checker's records (oracle/svt_oracle_coeffscan.c) are pinned on the REFERENCE's coder.
Two arithmetic-coder states of the reference (oracle/ref_harness_coeffscan.c inside oracle/_ref/libsvtref.so): one is fed the s16 coefficients through
the reference's own EncodeQuantizedCoefficients_generic / _SSE2 (Codec/EbEntropyCoding.c:1172 / :1716), the other the pre-scan records through the CABAC
loop a maintainer binds in (integration/svt_coeff_scan_consumer.h).  Same bytes written, same interval, same context models - block after block."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_encodepass_golden import ALL, load_case

vp, u32 = C.c_void_p, C.c_uint32


@pytest.fixture(scope="module")
def ref():
    lib = S.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libsvtref.so not built (needs /root/reference)")
    lib.svt_ref_cabac_new.restype, lib.svt_ref_cabac_new.argtypes = vp, [u32]
    lib.svt_ref_cabac_free.restype, lib.svt_ref_cabac_free.argtypes = None, [vp]
    lib.svt_ref_cabac_code_raw.restype, lib.svt_ref_cabac_code_raw.argtypes = None, [vp, u32, u32, u32, vp, u32, u32, u32, C.c_int]
    lib.svt_ref_cabac_code_scan.restype, lib.svt_ref_cabac_code_scan.argtypes = None, [vp, u32, u32, vp, vp, vp]
    lib.svt_ref_cabac_state.restype, lib.svt_ref_cabac_state.argtypes = u32, [vp, vp, u32]
    return lib


def scan_sigs(oracle):
    oracle.svt_oracle_coeff_scan_tu.restype = None
    oracle.svt_oracle_coeff_scan_tu.argtypes = [vp, u32, u32, u32, u32, C.c_int, u32, vp, vp, C.POINTER(u32), vp, C.POINTER(u32)]
    oracle.svt_oracle_coeff_scan_lcu.restype, oracle.svt_oracle_coeff_scan_lcu.argtypes = C.c_int, [vp, vp, vp, vp, vp]


def state_of(ref, h):
    buf = np.zeros(1 << 22, np.uint8)
    n = ref.svt_ref_cabac_state(h, buf.ctypes.data, buf.size)
    assert n
    return buf[:n].copy()


def random_block(rng, size, kind):
    c = np.zeros((size, size), np.int16)
    if kind == "dc":
        c[0, 0] = rng.choice([-3, -1, 1, 2, 5, 300])
    elif kind == "one":            # a single coefficient somewhere else: not the fast track
        c[rng.integers(0, size), rng.integers(0, size)] = rng.choice([-2, 1, 7])
        if c[0, 0]:
            c[0, 0], c[size - 1, size - 1] = 0, 1
    elif kind == "sparse":
        n = int(rng.integers(2, 9))
        c[rng.integers(0, size, n), rng.integers(0, size, n)] = rng.integers(-4, 5, n)
    elif kind == "lowpass":        # what a quantiser leaves: magnitudes falling with frequency
        yy, xx = np.mgrid[:size, :size]
        c[:] = (rng.integers(-40, 41, (size, size)) / (1 + (xx + yy) ** 1.5)).astype(np.int16)
    elif kind == "dense":
        c[:] = rng.integers(-3, 4, (size, size))
    elif kind == "big":
        c[:] = rng.choice([0, 0, 1, -1, 2, -700, 32767, -32768, 19], (size, size))
    elif kind == "corner":         # only the last sub-block of the scan
        c[size - 1, size - 1] = -1
        c[size - 2, size - 1] = 3
    return c


def test_prescan_records_code_the_reference_bytes_on_random_blocks(oracle, ref):
    scan_sigs(oracle)
    rng = np.random.default_rng(17)
    for asm_form in (0, 1):
        a, b = ref.svt_ref_cabac_new(99), ref.svt_ref_cabac_new(99)
        try:
            blocks = 0
            for rep in range(60):
                for size in (4, 8, 16, 32):
                    for kind in ("dc", "one", "sparse", "lowpass", "dense", "big", "corner"):
                        comp = int(rng.integers(0, 3))
                        if comp and size == 32:
                            comp = 0
                        typ = int(rng.choice([1, 2]))
                        mode = int(rng.integers(0, 35))
                        stride = 64 if comp == 0 else 32
                        plane = np.zeros((32, stride), np.int16)
                        blk = random_block(rng, size, kind)
                        plane[:size, :size] = blk
                        nz = int((blk != 0).sum())
                        if nz == 0:
                            continue
                        raw = plane.copy()
                        ref.svt_ref_cabac_code_raw(a, size, typ, mode, raw.ctypes.data, stride, comp, nz, asm_form)
                        tu = np.zeros(1, S.COEFF_SCAN_TU_DTYPE)
                        groups, levels = np.zeros(64, S.COEFF_SCAN_GROUP_DTYPE), np.zeros(1024, np.uint16)
                        ng, nl = u32(0), u32(0)
                        oracle.svt_oracle_coeff_scan_tu(plane.ctypes.data, stride, size, typ, mode, int(comp != 0), nz, tu.ctypes.data, groups.ctypes.data,
                                                        C.byref(ng), levels.ctypes.data, C.byref(nl))
                        assert nl.value == nz and (kind != "dc" or tu[0]["dc_only"])
                        ref.svt_ref_cabac_code_scan(b, size, comp, tu.ctypes.data, groups.ctypes.data, levels.ctypes.data)
                        blocks += 1
                if rep % 10 == 9:
                    sa, sb = state_of(ref, a), state_of(ref, b)
                    assert np.array_equal(sa, sb), (asm_form, rep, blocks, int(np.argmax(sa != sb)) if sa.size == sb.size else (sa.size, sb.size))
            assert blocks > 1500 and state_of(ref, a).size > 20000      # tens of kilobytes of coded coefficients
        finally:
            ref.svt_ref_cabac_free(a), ref.svt_ref_cabac_free(b)


CASES = [c for c in ALL if not c.startswith(("dlf_", "sao_"))]


@pytest.mark.parametrize("name", CASES)
def test_prescan_of_recorded_lcus_codes_the_reference_bytes(oracle, ref, name):
    """every LCU of the recorded encodes (I / P / B, 8- and 10-bit records, 64x64 inter units, skipped units): the LCU-level records, block by block in
    the entropy coder's order (unit, then Y / Cb / Cr), against the reference's coder on the coefficient planes"""
    scan_sigs(oracle)
    g, w, h = load_case(name)
    a, b = ref.svt_ref_cabac_new(5), ref.svt_ref_cabac_new(5)
    coded = 0
    try:
        for k in range(len(g["work"])):
            work, res = np.ascontiguousarray(g["work"][k:k + 1]), np.ascontiguousarray(g["result"][k:k + 1])
            lcu = np.zeros(1, S.COEFF_SCAN_LCU_DTYPE)
            groups, levels = np.zeros(384, S.COEFF_SCAN_GROUP_DTYPE), np.zeros(6144, np.uint16)
            n = oracle.svt_oracle_coeff_scan_lcu(work.ctypes.data, res.ctypes.data, lcu.ctypes.data, groups.ctypes.data, levels.ctypes.data)
            wk, rs = work[0], res[0]
            ncu = int(wk["num_cus"])
            big = ncu == 1 and int(wk["cu"][0]["size"]) == 64
            seen = 0
            for c in (range(1, 5) if big else range(ncu)):
                cu = wk["cu"][0 if big else c]
                size = 32 if big else int(cu["size"])
                x, y = (32 * ((c - 1) & 1), 32 * ((c - 1) >> 1)) if big else (int(cu["x"]), int(cu["y"]))
                for p in range(3):
                    tu = lcu[0]["tu"][p][c:c + 1]
                    if not rs["cu"][c]["cbf"][p]:
                        assert tu[0]["last_scan_set"] == -1
                        continue
                    ts = size if p == 0 else (4 if size == 8 else size // 2)
                    plane = (rs["coeff_y"].reshape(64, 64) if p == 0 else rs[("coeff_cb", "coeff_cr")[p - 1]].reshape(32, 32)).copy()
                    sub = np.ascontiguousarray(plane[(y >> (p > 0)):, (x >> (p > 0)):][:ts, :ts])
                    nzc = int(rs["cu"][c]["nz"][p])
                    assert nzc == int((sub != 0).sum()), (name, k, c, p)
                    st = 64 if p == 0 else 32
                    full = np.zeros((32, st), np.int16)
                    full[:ts, :ts] = sub
                    ref.svt_ref_cabac_code_raw(a, ts, int(cu["pred_mode"]), int(cu["intra_luma_mode"]), full.ctypes.data, st, p, nzc, 0)
                    ref.svt_ref_cabac_code_scan(b, ts, p, np.ascontiguousarray(tu).ctypes.data, groups.ctypes.data, levels.ctypes.data)
                    seen += 1
            assert seen == n
            assert int(lcu[0]["levels"]) == sum(int(rs["cu"][c]["nz"][p]) for c in (range(1, 5) if big else range(ncu)) for p in range(3) if rs["cu"][c]["cbf"][p])
            coded += seen
        sa, sb = state_of(ref, a), state_of(ref, b)
        assert np.array_equal(sa, sb), (name, coded)
        assert coded > 0
    finally:
        ref.svt_ref_cabac_free(a), ref.svt_ref_cabac_free(b)


def test_layouts():
    assert S.COEFF_SCAN_LCU_DTYPE.itemsize == 1552
