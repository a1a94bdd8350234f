"""CPU-only: ComputeDecimatedZzSad restatement (oracle/svt_oracle_zz.c).  The reference function is `static`, so it is
pinned through the two reference leafs it is made of - Decimation2D and FastLoop_NxMSadKernel called in libsvtref.so -
and the BEA_CLASS_* ladders quoted from Codec/EbDefinitions.h:1087-1107."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S

ref = S.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libsvtref.so not built")
ZZ = np.dtype([("sad", "<u4"), ("zz_cost", "u1"), ("non_moving_index", "u1"), ("pad", "u1", 2)])
u32, vp = C.c_uint32, C.c_void_p


def oracle_zz(oracle, cur, prev):
    h, w = cur.shape
    n = S.lcu_count(w, h)
    out = np.zeros(n, ZZ)
    oracle.svt_oracle_zz_sad_picture.argtypes = [vp, vp, u32, u32, u32, vp]
    oracle.svt_oracle_zz_sad_picture.restype = None
    oracle.svt_oracle_zz_sad_picture(np.ascontiguousarray(cur).ctypes.data, np.ascontiguousarray(prev).ctypes.data, w, w, h,
                                     out.ctypes.data)
    return out


@pytest.mark.parametrize("kind,w,h", [("motion", 416, 240), ("flat", 320, 256), ("noise", 328, 264)])
def test_zz_matches_reference_leafs(oracle, kind, w, h):
    ref.FastLoop_NxMSadKernel.restype = u32
    cur, prev = S.gen_luma(kind, w, h, 3, 7), S.gen_luma(kind, w, h, 2, 7)
    if kind == "motion":  # make some LCUs static so every class of both ladders occurs
        cur = cur.copy()
        cur[:, :192] = prev[:, :192]
        cur[:64, 64:128] = np.clip(prev[:64, 64:128].astype(int) + (np.arange(64)[None, :] % 5 == 0) * 9, 0, 255)
    got = oracle_zz(oracle, cur, prev)
    wl = (w + 63) // 64
    for l in range(len(got)):
        ox, oy = (l % wl) * 64, (l // wl) * 64
        lw, lh = min(64, w - ox), min(64, h - oy)
        if lw == 64 and lh == 64:
            c16, p16 = np.zeros((16, 16), np.uint8), np.zeros((16, 16), np.uint8)
            for plane, dst in ((cur, c16), (prev, p16)):
                blk = np.ascontiguousarray(plane[oy:oy + 64, ox:ox + 64])
                ref.Decimation2D(vp(blk.ctypes.data), u32(64), u32(64), u32(64), vp(dst.ctypes.data), u32(16), u32(4))
            sad = ref.FastLoop_NxMSadKernel(vp(c16.ctypes.data), u32(16), vp(p16.ctypes.data), u32(16), u32(16), u32(16))
            zz = 0 if sad < 256 else 3 if sad < 512 else 10 if sad < 1024 else 20 if sad < 2048 else 30
        else:
            sad, zz = 0xFFFFFFFF, 0xFF
        area = (lw >> 2) * (lh >> 2)
        nm = 0 if sad < area * 2 else 10 if sad < area * 4 else 20 if sad < area * 8 else 30
        assert (int(got[l]["sad"]), int(got[l]["zz_cost"]), int(got[l]["non_moving_index"])) == (sad, zz, nm), l
    if kind == "motion":
        assert len(set(got["zz_cost"].tolist())) >= 3
