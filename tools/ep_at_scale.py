import sys, os, tempfile
sys.path.insert(0, "tools")
import encoder_fps as E
for cfg in ("cfg3", "cfg4"):
    with tempfile.TemporaryDirectory() as td:
        rp = os.path.join(td, "r.txt")
        r = E.measure(cfg, frames=33, unique=16, hip_env={"SVT_HOOK_ENCODEPASS": "1", "SVT_HOOK_REPORT": rp})
        print(cfg, "ref %.1f hip %.1f identical %s" % (r["reference"]["fps"], r["hip"]["fps"], r["bitstream_identical"]))
        print([l.strip() for l in open(rp) if "encode pass" in l])
