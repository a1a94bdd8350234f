"""CPU-only: oracle/svt_oracle_mcp.c against the REFERENCE's own dispatch tables (Codec/EbMcpTables.c), called
through oracle/_ref/libsvtref.so: every luma position (16), every chroma position (64), uni and bi (raw) forms,
8- and 16-bit, slot 0 (C_DEFAULT).  Slot 1 (SSSE3/SSE2 intrinsics) is cross-checked for the uni-prediction (final
sample) tables only: its raw int16 intermediates use a private 8-column-strip layout shared with BiPredClipping_SSSE3,
so they are not comparable element by element (the product implements the C_DEFAULT convention)."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
u32, vp, i32 = C.c_uint32, C.c_void_p, C.c_int32

UNI = C.CFUNCTYPE(None, vp, u32, vp, u32, u32, u32, vp)                 # InterpolationFilterNew
RAW = C.CFUNCTYPE(None, vp, u32, vp, u32, u32, vp)                      # InterpolationFilterOutRaw
CUNI = C.CFUNCTYPE(None, vp, u32, vp, u32, u32, u32, vp, u32, u32)      # ChromaFilterNew
CRAW = C.CFUNCTYPE(None, vp, u32, vp, u32, u32, vp, u32, u32)           # ChromaFilterOutRaw


def table(name, n, proto, slot=0):
    arr = (C.c_void_p * (2 * n)).in_dll(ref, name)
    return [proto(arr[slot * n + i]) for i in range(n)]


def decl(oracle):
    oracle.svt_oracle_mcp.argtypes = [C.c_int, C.c_int, C.c_int, u32, u32, vp, u32, vp, u32, u32, u32]
    oracle.svt_oracle_mcp.restype = None
    oracle.svt_oracle_BiPredClipping.argtypes = [C.c_int, u32, u32, vp, vp, vp, u32, i32]
    oracle.svt_oracle_BiPredClipping.restype = None


def refplane(rng, bps, extreme):
    hi = 256 if bps == 1 else 1024
    if extreme:  # worst-case magnitudes: checker of 0 / max
        yy, xx = np.mgrid[0:96, 0:112]
        a = ((yy + xx + rng.integers(0, 2)) & 1) * (hi - 1)
    else:
        a = rng.integers(0, hi, (96, 112))
    return a.astype(np.uint8 if bps == 1 else np.uint16)


SIZES = [(8, 8), (16, 8), (8, 32), (32, 32), (64, 64), (24, 16)]


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("slot", [0, 1])
def test_luma_tables(oracle, bps, slot):
    decl(oracle)
    uni = table("uniPredLumaIFFunctionPtrArrayNew" if bps == 1 else "uniPredLuma16bitIFFunctionPtrArray", 16, UNI, slot)
    raw = table("biPredLumaIFFunctionPtrArrayNew" if bps == 1 else "biPredLumaIFFunctionPtrArrayNew16bit", 16, RAW, slot)
    rng = np.random.default_rng(bps)
    dt = np.uint8 if bps == 1 else np.uint16
    for pos in range(16):
        fx, fy = pos & 3, pos >> 2
        for k, (w, h) in enumerate(SIZES):
            plane = refplane(rng, bps, k == 1)
            base = plane.ctypes.data + (16 * 112 + 16) * bps
            tmp = np.zeros((h + 8) * w + 64, np.int16)
            want, got = np.full((h, 80), 7, dt), np.full((h, 80), 7, dt)
            uni[pos](base, 112, want.ctypes.data, 80, w, h, tmp.ctypes.data)
            oracle.svt_oracle_mcp(bps, 0, 0, fx, fy, base, 112, got.ctypes.data, 80, w, h)
            assert np.array_equal(want, got), ("uni", pos, w, h)
            if slot:
                continue
            want, got = np.full(h * w, 7, np.int16), np.full(h * w, 7, np.int16)
            raw[pos](base, 112, want.ctypes.data, w, h, tmp.ctypes.data)
            oracle.svt_oracle_mcp(bps, 0, 1, fx, fy, base, 112, got.ctypes.data, 0, w, h)
            assert np.array_equal(want, got), ("raw", pos, w, h)


@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("slot", [0, 1])
def test_chroma_tables(oracle, bps, slot):
    decl(oracle)
    uni = table("uniPredChromaIFFunctionPtrArrayNew" if bps == 1 else "uniPredChromaIFFunctionPtrArrayNew16bit", 64, CUNI, slot)
    raw = table("biPredChromaIFFunctionPtrArrayNew" if bps == 1 else "biPredChromaIFFunctionPtrArrayNew16bit", 64, CRAW, slot)
    rng = np.random.default_rng(10 + bps)
    dt = np.uint8 if bps == 1 else np.uint16
    for pos in range(64):
        fx, fy = pos & 7, pos >> 3
        for k, (w, h) in enumerate([(4, 4), (8, 4), (16, 16), (32, 32), (4, 16)]):
            plane = refplane(rng, bps, k == 1)
            base = plane.ctypes.data + (16 * 112 + 16) * bps
            tmp = np.zeros((h + 4) * w + 64, np.int16)
            want, got = np.full((h, 48), 7, dt), np.full((h, 48), 7, dt)
            uni[pos](base, 112, want.ctypes.data, 48, w, h, tmp.ctypes.data, fx, fy)
            oracle.svt_oracle_mcp(bps, 1, 0, fx, fy, base, 112, got.ctypes.data, 48, w, h)
            assert np.array_equal(want, got), ("uni", pos, w, h)
            if slot:
                continue
            want, got = np.full(h * w, 7, np.int16), np.full(h * w, 7, np.int16)
            raw[pos](base, 112, want.ctypes.data, w, h, tmp.ctypes.data, fx, fy)
            oracle.svt_oracle_mcp(bps, 1, 1, fx, fy, base, 112, got.ctypes.data, 0, w, h)
            assert np.array_equal(want, got), ("raw", pos, w, h)


def test_bipred_clipping(oracle):
    decl(oracle)
    rng = np.random.default_rng(3)
    for w, h in SIZES:
        l0 = rng.integers(-8192, 8192, h * w).astype(np.int16)
        l1 = rng.integers(-8192, 8192, h * w).astype(np.int16)
        for offset in (64 + 16384, 64):  # Offset5 (luma) / ChromaOffset5
            want, got = np.full((h, 80), 7, np.uint8), np.full((h, 80), 7, np.uint8)
            ref.BiPredClipping(u32(w), u32(h), vp(l0.ctypes.data), vp(l1.ctypes.data), vp(want.ctypes.data), u32(80), i32(offset))
            oracle.svt_oracle_BiPredClipping(1, w, h, l0.ctypes.data, l1.ctypes.data, got.ctypes.data, 80, offset)
            assert np.array_equal(want, got)
        want, got = np.full((h, 80), 7, np.uint16), np.full((h, 80), 7, np.uint16)
        ref.BiPredClipping16bit(u32(w), u32(h), vp(l0.ctypes.data), vp(l1.ctypes.data), vp(want.ctypes.data), u32(80))
        oracle.svt_oracle_BiPredClipping(2, w, h, l0.ctypes.data, l1.ctypes.data, got.ctypes.data, 80, 0)
        assert np.array_equal(want, got)
