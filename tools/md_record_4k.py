#!/usr/bin/env python3
"""Records the inputs and the reference's decisions of the open-loop B pictures of a 5-picture 4K encode at BASELINE configs[2]'s preset (encMode 7, random access) with the
prebuilt reference (oracle/_ref, SVT_REF_MD_DUMP) -> tools/_fx/md4k.npz (git-ignored; it travels to the GPU box with the snapshot, so the box does not spend a minute
recording it on every call).  Pictures: [0] layer 2 non-reference, [1] layer 1 reference (CHROMA_MODE_FULL LCUs), [2] layer 2 non-reference."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def record(path, w=3840, h=2160):
    import md_bench
    g = md_bench.record_inter(w, h, 7, frames=5, kind="motion", levels=2, ref=None)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **g)
    return g


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_fx", "md4k.npz")
    g = record(out)
    print(out, [int(p["temporal_layer"]) for p in g["pic"]], os.path.getsize(out) >> 20, "MB")
