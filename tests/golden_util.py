"""Load the committed ME golden fixtures (tests/golden/me_*.npz)."""
import glob
import os

import numpy as np

import svtlib as S


def golden_cases():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "me_*.npz")))


def load_case(name):
    z = np.load(os.path.join(S.GOLDEN_DIR, "me_%s.npz" % name))
    kind, w, h, n, seed = z["clip"]
    rows = [int(r) for r in z["rows_kept"]] if "rows_kept" in z.files else None  # big fixtures keep some LCU rows only
    return dict(kind=str(kind), w=int(w), h=int(h), n=int(n), seed=int(seed), meta=z["meta"],
                params=z["params"].view(S.ME_PARAMS_DTYPE), results=z["results"].view(S.ME_LCU_DTYPE), rows_kept=rows)


def kept_lcus(g):
    """LCU indices a fixture holds results for (None = all)."""
    if g["rows_kept"] is None:
        return None
    wl = (g["w"] + 63) // 64
    return [r * wl + c for r in g["rows_kept"] for c in range(wl)]
