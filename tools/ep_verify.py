#!/usr/bin/env python3
"""Encode-pass binding in verification mode (needs the GPU): the hooked encoder runs with SVT_HOOK_ENCODEPASS=1 and
SVT_HOOK_ENCODEPASS_VERIFY=1 - every LCU the device call covers is encoded on the device AND by the reference's own EncodePass, and the
two outcomes (TransformUnit_t flags, quantised coefficients, and with `-dlf 1 -sao 0` the un-deblocked reconstruction) are compared
unit by unit inside the encoder (integration/svt_hook_encdec.c:verify_lcu).  Prints the binding's report and the first mismatches.
usage: python tools/ep_verify.py kind width height frames [encoder arguments ...]      kind: motion | noise | flat [+ "10" / "10c"]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402


def main():
    kind, w, h, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    args = sys.argv[5:]
    app = os.path.join(ROOT, "integration", "_build", "SvtHevcEncApp_hip")
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "clip.yuv")
        if kind.endswith("10c"):
            S.write_clip10_compressed(yuv, kind[:-3], w, h, n, 7)
        elif kind.endswith("10"):
            S.write_clip10(yuv, kind[:-2], w, h, n, 7)
        else:
            S.write_clip(yuv, kind, w, h, n, 7)
        rep = os.path.join(td, "report.txt")
        r = subprocess.run([app, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-b", os.path.join(td, "out.265")] +
                           ([] if "-asm" in args else ["-asm", "1"]) + ([] if "-q" in args else ["-q", "32"]) + args, capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, SVT_HOOK_ENCODEPASS="1", SVT_HOOK_ENCODEPASS_VERIFY="1", SVT_HOOK_REPORT=rep))
        lines = [ln for ln in r.stderr.splitlines() if "VERIFY" in ln]
        print("\n".join(lines[:40]))
        print("exit code", r.returncode, "; mismatch lines", len(lines))
        if r.returncode:
            print(r.stdout[-1500:], r.stderr[-1500:])
        if os.path.exists(rep):
            print("\n".join(ln for ln in open(rep).read().splitlines() if "encode pass" in ln))
        return 1 if (r.returncode or lines) else 0


if __name__ == "__main__":
    sys.exit(main())
