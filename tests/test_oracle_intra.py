"""CPU-only: the oracle's intra-prediction restatement vs the reference's 24 C_DEFAULT kernels."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
u32, i32, vp = C.c_uint32, C.c_int32, C.c_void_p

# mode id -> (8-bit symbol, 16-bit symbol, takes angle)
KERNELS = {
    0: ("IntraModeVerticalLuma", "IntraModeVerticalLuma16bit", False),
    1: ("IntraModeVerticalChroma", "IntraModeVerticalChroma16bit", False),
    2: ("IntraModeHorizontalLuma", "IntraModeHorizontalLuma16bit", False),
    3: ("IntraModeHorizontalChroma", "IntraModeHorizontalChroma16bit", False),
    4: ("IntraModeDCLuma", "IntraModeDCLuma16bit", False),
    5: ("IntraModeDCChroma", "IntraModeDCChroma16bit", False),
    6: ("IntraModePlanar", "IntraModePlanar16bit", False),
    7: ("IntraModeAngular_34", "IntraModeAngular16bit_34", False),
    8: ("IntraModeAngular_18", "IntraModeAngular16bit_18", False),
    9: ("IntraModeAngular_2", "IntraModeAngular16bit_2", False),
    10: ("IntraModeAngular_Vertical_Kernel", "IntraModeAngular16bit_Vertical_Kernel", True),
    11: ("IntraModeAngular_Horizontal_Kernel", "IntraModeAngular16bit_Horizontal_Kernel", True),
}
ANGLES = [32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32]


def make_refs(size, bps, seed):
    rng = np.random.default_rng(seed)
    hi = 256 if bps == 1 else 1024
    # generous margins on both sides: the generic angular kernels index refSampMain[-size .. 2*size+2]
    return rng.integers(0, hi, size=8 * size + 16).astype(np.uint8 if bps == 1 else np.uint16)


@pytest.mark.parametrize("mode", sorted(KERNELS))
@pytest.mark.parametrize("bps", [1, 2])
@pytest.mark.parametrize("size", [4, 8, 16, 32, 64])
def test_intra_kernels(oracle, mode, bps, size):
    oracle.svt_oracle_IntraPred.argtypes = [C.c_int, C.c_int, u32, vp, vp, u32, C.c_int, i32]
    name = KERNELS[mode][bps - 1]
    fn = getattr(ref, name)
    dt = np.uint8 if bps == 1 else np.uint16
    for skip in (0, 1):
        for angle in (ANGLES if KERNELS[mode][2] else [0]):
            refs = make_refs(size, bps, size * 7 + mode + bps)
            base = refs.ctypes.data + (2 * size + 4) * bps if KERNELS[mode][2] else refs.ctypes.data
            want = np.full((size, size + 8), 77, dt)
            got = want.copy()
            if KERNELS[mode][2]:
                fn(u32(size), vp(base), vp(want.ctypes.data), u32(size + 8), C.c_ubyte(skip), i32(angle))
            else:
                fn(u32(size), vp(base), vp(want.ctypes.data), u32(size + 8), C.c_ubyte(skip))
            oracle.svt_oracle_IntraPred(mode, bps, size, base, got.ctypes.data, size + 8, skip, angle)
            assert np.array_equal(got, want), (name, size, skip, angle)
