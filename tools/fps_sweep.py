"""Encoder fps of the hooked encoder against the reference over run lengths (start-up vs steady state), with the hook's front-half
timeline.  usage (GPU box): python tools/fps_sweep.py [cfg:frames[,frames...]] ..."""
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import encoder_fps as E

specs = sys.argv[1:] or ["cfg3:96", "cfg2:256"]
for spec in specs:
    cfg, ns = spec.split(":")
    for n in (int(v) for v in ns.split(",")):
        with tempfile.TemporaryDirectory() as td:
            rp = os.path.join(td, "report.txt")
            r = E.measure(cfg, frames=n, unique=16, hip_env={"SVT_HOOK_REPORT": rp, "SVT_HOOK_VERBOSE": "1"})
            print(cfg, n, "ref fps %.1f wall %.2f | hip fps %.1f wall %.2f | identical %s" % (r["reference"]["fps"], r["reference"]["wall_s"], r["hip"]["fps"],
                                                                                             r["hip"]["wall_s"], r["bitstream_identical"]), flush=True)
            for line in open(rp):
                if "timeline" in line or "encode pass" in line or "first pictures" in line:
                    print("   ", line.strip(), flush=True)
