"""CPU-only: pins oracle/svt_oracle_mcp.c:svt_oracle_inter_pu (position clamp, integer / fractional split for luma and
chroma, interpolation of the three planes, bi-prediction average) against records of real EncodePassInterPrediction calls of
the reference's encode pass and the padded reference pictures they read (tests/golden/inter_*.npz, made by
tests/golden/make_inter_golden.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "inter_*.npz")))
CASES16 = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "inter16_*.npz")))  # 10-bit encodes

JOB = np.dtype([("mv", "<i2", (2, 2)), ("pu_x", "<u2"), ("pu_y", "<u2"), ("pu_w", "u1"), ("pu_h", "u1"), ("pred_dir", "u1"), ("pad", "u1"),
                ("dst_off_y", "<i4"), ("dst_off_c", "<i4")])


class RefPicture(C.Structure):
    _fields_ = [("d_y", C.c_void_p), ("d_cb", C.c_void_p), ("d_cr", C.c_void_p), ("strideY", C.c_uint32), ("strideC", C.c_uint32),
                ("originX", C.c_uint32), ("originY", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


def load_inter_case(name, hbd=False):
    g = dict(np.load(os.path.join(S.GOLDEN_DIR, ("inter16_%s.npz" if hbd else "inter_%s.npz") % name)))
    n = g["pu_w"].astype(np.int64) * g["pu_h"].astype(np.int64)
    g["off_y"] = np.concatenate([[0], np.cumsum(n)])
    g["off_c"] = np.concatenate([[0], np.cumsum(n // 4)])
    return g


def ref_struct(g, pid, ptrs):
    h = g["pic%d_hdr" % pid][0]
    r = RefPicture()
    r.d_y, r.d_cb, r.d_cr = ptrs
    r.strideY, r.strideC, r.originX, r.originY = int(h["strideY"]), int(h["strideC"]), int(h["originX"]), int(h["originY"])
    r.width, r.height = int(h["width"]), int(h["height"])
    return r


def job_of(g, i):
    j = np.zeros(1, JOB)
    for k in ("mv", "pu_x", "pu_y", "pu_w", "pu_h", "pred_dir"):
        j[k] = g[k][i]
    return j


def want_of(g, i):
    w, h = int(g["pu_w"][i]), int(g["pu_h"][i])
    a, b, c, d = int(g["off_y"][i]), int(g["off_y"][i + 1]), int(g["off_c"][i]), int(g["off_c"][i + 1])
    return g["pred_y"][a:b].reshape(h, w), g["pred_cb"][c:d].reshape(h // 2, w // 2), g["pred_cr"][c:d].reshape(h // 2, w // 2)


def test_struct_sizes():
    assert JOB.itemsize == 24 and C.sizeof(RefPicture) == 48


def test_have_cases():
    assert len(CASES) >= 4 and len(CASES16) == 2


@pytest.mark.parametrize("name,hbd", [(n, False) for n in CASES] + [(n, True) for n in CASES16])
def test_inter_pu_oracle_matches_reference(oracle, name, hbd):
    g = load_inter_case(name, hbd)
    fn = oracle.svt_oracle_inter_pu16bit if hbd else oracle.svt_oracle_inter_pu
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    fn.restype = None
    assert g["pred_y"].dtype == (np.uint16 if hbd else np.uint8) and (not hbd or int(g["pred_y"].max()) > 255)
    planes = {int(p): [np.ascontiguousarray(g["pic%d_%s" % (p, c)]) for c in ("y", "cb", "cr")] for p in g["pic_ids"]}
    refs = {p: ref_struct(g, p, [a.ctypes.data for a in planes[p]]) for p in planes}
    seen = set()
    for i in range(len(g["pu_w"])):
        want = want_of(g, i)
        got = [np.zeros_like(w) for w in want]
        j = job_of(g, i)
        r0, r1 = (refs.get(int(v)) for v in g["ref_id"][i])
        fn(j.ctypes.data, C.addressof(r0) if r0 else None, C.addressof(r1) if r1 else None, got[0].ctypes.data,
           got[0].shape[1], got[1].ctypes.data, got[2].ctypes.data, got[1].shape[1])
        for p in range(3):
            assert np.array_equal(got[p], want[p]), (name, i, p, g["mv"][i].tolist(), int(g["pred_dir"][i]),
                                                     np.argwhere(got[p] != want[p])[:4].tolist())
        seen.add((int(g["pred_dir"][i]), int(g["mv"][i][0][0]) & 3, int(g["mv"][i][0][1]) & 3))
    assert len(seen) >= (2 if hbd else 3)
