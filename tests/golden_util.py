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
    return dict(kind=str(kind), w=int(w), h=int(h), n=int(n), seed=int(seed), meta=z["meta"],
                params=z["params"].view(S.ME_PARAMS_DTYPE), results=z["results"].view(S.ME_LCU_DTYPE))
