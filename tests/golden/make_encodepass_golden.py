#!/usr/bin/env python3
"""Generate the encode-pass (LCU) golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref with SVT_REF_ENCODEPASS_DUMP set and the loop filters off (-dlf 1 -sao 0, so the reconstruction
buffer still holds the un-deblocked samples when EncodePass returns): the --wrap interposer oracle/ref_harness_encodepass_dump.c
records, for every LCU whose coding units are all intra 2Nx2N, the input contract (coding-unit list + source samples =
SvtAmdLcuWork) and what the reference's EncodePass produced (TransformUnit_t flags, quantizedCoeff, reconstruction =
SvtAmdLcuResult).  Stored as tests/golden/encodepass_<name>.npz.  Needs /root/reference.  Usage: make_encodepass_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

CASES = {
    # name -> (clip, w, h, frames, seed, args): all-intra pictures; 416x240 has a partial right column and bottom row of LCUs
    "i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "30"]),
    # worst-case residual energy at a low qp: every unit size, large levels, DC-only units rare
    "i_noise_200x136_m6": ("noise", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "22"]),
    # smooth content at a high qp: many units without coefficients and many DC-only units (EncodeInvTransform's shortcut)
    "i_motion_320x192_m7_q44": ("motion", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "44"]),
    # 10-bit encodes (EncodePass with is16bit: EncodeLoop16bit, 16-bit intra prediction, quantiser at qp + 12): "10" in the clip kind
    "i10_motion_416x240_m9": ("motion10", 416, 240, 1, 7, ["-encMode", "9", "-intra-period", "0", "-q", "30", "-bit-depth", "10"]),
    "i10_noise_200x136_m6": ("noise10", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "24", "-bit-depth", "10"]),
    "i10_motion_320x192_m7_q45": ("motion10", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "45", "-bit-depth", "10"]),
    # deblocking ON (the encoder's default), SAO off: the encoder's reconstruction output is then the deblocked picture, the
    # fixture for encode pass -> boundary strengths -> deblocking chained on the device ("dlf_" prefix: `recon_*` planes are kept,
    # the per-LCU `rec_*` of the records are partly deblocked and not compared)
    "dlf_i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "34"]),
    "dlf_i_noise_200x136_m6": ("noise", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "38"]),
    "dlf_i10_motion_320x192_m7": ("motion10", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "36", "-bit-depth", "10"]),
}


def run_case(name):
    kind, w, h, n, seed, args = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "ep.dump")
        if kind.endswith("10"):
            S.write_clip10(yuv, kind[:-2], w, h, n, seed)
        else:
            S.write_clip(yuv, kind, w, h, n, seed)
        dlf = name.startswith("dlf_")
        rec_out = os.path.join(td, "rec.yuv")
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-sao", "0",
               "-b", os.path.join(td, "out.265")] + ([] if dlf else ["-dlf", "1"]) + (["-o", rec_out] if dlf else []) + args
        subprocess.run(cmd, env=dict(os.environ, SVT_REF_ENCODEPASS_DUMP=dump), check=True, stdout=subprocess.DEVNULL)
        rdt = S.EP_RECORD16_DTYPE if kind.endswith("10") else S.EP_RECORD_DTYPE
        recs = np.fromfile(dump, dtype=rdt)
        rec_raw = open(rec_out, "rb").read() if dlf else b""
    assert len(recs) and (recs["record_size"] == rdt.itemsize).all(), (len(recs), rdt.itemsize)
    assert (recs["dlf_off"] == (0 if dlf else 1)).all()
    extra = {}
    if dlf:   # the encoder's reconstruction output: n frames, 4:2:0, 8-bit samples or 16-bit little-endian for 10-bit encodes
        sdt = np.dtype("<u2") if kind.endswith("10") else np.uint8
        raw = np.frombuffer(rec_raw, sdt)
        fs = w * h * 3 // 2
        assert raw.size == n * fs, (raw.size, n, fs)
        extra = {"recon_y": np.stack([raw[i * fs:i * fs + w * h].reshape(h, w) for i in range(n)]),
                 "recon_cb": np.stack([raw[i * fs + w * h:i * fs + w * h * 5 // 4].reshape(h // 2, w // 2) for i in range(n)]),
                 "recon_cr": np.stack([raw[i * fs + w * h * 5 // 4:(i + 1) * fs].reshape(h // 2, w // 2) for i in range(n)])}
    nl = S.lcu_count(w, h)
    order = np.lexsort((recs["lcu_index"], recs["picture_number"]))
    recs = recs[order]
    assert len(recs) == nl * n, "every LCU of an all-intra clip must be recorded (%d of %d)" % (len(recs), nl * n)
    path = os.path.join(S.GOLDEN_DIR, "encodepass_%s.npz" % name)
    np.savez_compressed(path, clip=np.array([kind, str(w), str(h), str(n), str(seed)]), enc_args=np.array(args),
                        picture_number=recs["picture_number"], lcu_index=recs["lcu_index"], work=recs["work"], result=recs["result"], **extra)
    sizes, cnt = np.unique(np.concatenate([r["work"]["cu"]["size"][:r["work"]["num_cus"]] for r in recs]), return_counts=True)
    cbf = np.concatenate([r["result"]["cu"]["cbf"][:r["work"]["num_cus"]] for r in recs])
    odc = np.concatenate([r["result"]["cu"]["only_dc"][:r["work"]["num_cus"]] for r in recs])
    print("%-28s %d LCUs -> %s (%.0f KiB); unit sizes %s; cbf set %s of %d; DC-only %s" %
          (name, len(recs), os.path.basename(path), os.path.getsize(path) / 1024, dict(zip(sizes.tolist(), cnt.tolist())),
           cbf.sum(axis=0).tolist(), len(cbf), odc.sum(axis=0).tolist()))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
