"""CPU-only: pins oracle/svt_oracle_ois.c against before/after dumps of the REFERENCE's OpenLoopIntraSearchLcu
recorded inside real encoder runs (tests/golden/ois_*.npz, made by tests/golden/make_ois_golden.py)."""
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "ois_*.npz")))


def load_ois_case(name):
    g = np.load(os.path.join(S.GOLDEN_DIR, "ois_%s.npz" % name))
    kind, w, h, n, seed = g["clip"]
    return g, kind, int(w), int(h), int(seed)


def kept_lcus(g, w, h):
    """LCU indices the fixture holds records for (big fixtures keep some LCU rows only)."""
    if "rows_kept" not in g.files:
        return np.arange(S.lcu_count(w, h))
    wl = (w + 63) // 64
    return np.concatenate([np.arange(int(r) * wl, (int(r) + 1) * wl) for r in g["rows_kept"]])


def me_like(me_sad, kept=None, n=None):
    """ME_LCU_DTYPE array carrying only distortion[0] (all OIS reads of the ME results); LCUs outside `kept` get zeros
    (their results are not compared)."""
    me = np.zeros(len(me_sad) if n is None else n, S.ME_LCU_DTYPE)
    if kept is None:
        me["pu"]["distortion"][:, :, 0] = me_sad
    else:
        me["pu"]["distortion"][kept, :, 0] = me_sad
    return me


def test_have_cases():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_ois_oracle_matches_reference(oracle, name):
    g, kind, w, h, seed = load_ois_case(name)
    seen = set()
    for i, (pn, slice_type, enc_mode) in enumerate(g["meta"]):
        luma = S.gen_luma(kind, w, h, int(pn), seed)
        params = S.ois_params_from_record(g["params"][i])
        kept = kept_lcus(g, w, h)
        me = me_like(g["me_sad"][i], kept, S.lcu_count(w, h)) if slice_type != 2 else None
        out = S.oracle_ois_picture(oracle, params, luma, me)[kept]
        got = S.ois_apply(g["before"][i], out)
        want = g["after"][i]
        bad = np.nonzero((got["candidate"] != want["candidate"]).any(axis=(1, 2)) | (got["total"] != want["total"]).any(axis=1))[0]
        assert len(bad) == 0, (name, int(pn), bad[:5])
        # the call really wrote something everywhere (guards against a vacuous mask)
        assert (out["candidate"][:, 1:21, 0] & (S.OIS_W_DIST | S.OIS_W_VALID)).any(axis=1).all()
        seen.add((int(slice_type), int(params.ois_kernel_level), int(params.limit_ois_to_dc_mode)))
    assert len(seen) >= 1


def test_paths_covered():
    """Across the fixtures: I slices, the OIS-point path, the DC-only path and the 35-mode kernel-level path."""
    seen = set()
    for name in CASES:
        g = np.load(os.path.join(S.GOLDEN_DIR, "ois_%s.npz" % name))
        for i, (pn, st, em) in enumerate(g["meta"]):
            p = g["params"][i]
            seen.add((int(st) == 2, int(p["ois_kernel_level"]), int(p["limit_ois_to_dc_mode"])))
    assert {(True, 0, 0), (False, 0, 0), (False, 0, 1), (False, 1, 0)} <= seen or \
        {(True, 1, 0), (False, 0, 0), (False, 0, 1), (False, 1, 0)} <= seen, seen
