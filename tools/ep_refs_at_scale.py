"""BASELINE configs[2] (4K, encMode 7, random access) with the encode pass AND the completion of reference pictures on the device
(SVT_HOOK_ENCODEPASS=1 SVT_HOOK_ENCODEPASS_REFS=verify): report lines + bitstream identity.  usage: python tools/ep_refs_at_scale.py [cfg] [frames]"""
import os
import sys
import tempfile
sys.path.insert(0, "tools")
import encoder_fps as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 17
with tempfile.TemporaryDirectory() as td:
    rp = os.path.join(td, "r.txt")
    log = os.path.join(td, "err.txt")
    os.environ["SVT_TOOLS_STDERR_TO"] = log
    r = E.measure(cfg, frames=frames, unique=min(frames, 16), hip_env={"SVT_HOOK_ENCODEPASS": "1", "SVT_HOOK_ENCODEPASS_REFS": "verify", "SVT_HOOK_REPORT": rp})
    print(cfg, frames, "frames: ref %.1f fps, hooked %.1f fps, bitstream identical %s" % (r["reference"]["fps"], r["hip"]["fps"], r["bitstream_identical"]))
    for l in open(rp):
        if "encode pass" in l or "reference pictures" in l:
            print("  ", l.strip())
    if os.path.exists(log):
        for l in open(log):
            if "REFVERIFY" in l:
                print("  ", l.strip())
