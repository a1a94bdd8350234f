#!/usr/bin/env python3
"""GPU box: how many pictures' mode decision + encode pass share the MI355X.  One recorded 4K B picture of BASELINE configs[2] (the reference run here with the recording
harness on), `flights` host threads each repeating the device call on their own lane and picture object, for several launch widths (SVT_AMD_MD_GRID = workgroups per
picture; 0 = the library's choice).  usage: md_flights.py [grids, comma separated] [flights, comma separated] [ref|nonref]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import md_bench
import svtlib as S

grids = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,32,20").split(",")]
flights = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
ref = len(sys.argv) > 3 and sys.argv[3] == "ref"
g = md_bench.record_inter(3840, 2160, 7, frames=5, kind="motion", levels=2, ref=ref)
lib = S.load_product()
out = []
for gr in grids:
    if gr:
        os.environ["SVT_AMD_MD_GRID"] = str(gr)
    else:
        os.environ.pop("SVT_AMD_MD_GRID", None)
    for f in flights:
        r = md_bench.run_inter_flights(lib, g, f, reps=3)
        r["grid"] = gr
        out.append(r)
        print(json.dumps(r), flush=True)
print(json.dumps({"picture": "4K encMode 7 %s B picture, mode decision + merge / skip decisions + encode pass, host-array ABI" % ("reference (CHROMA_MODE_FULL)" if ref else "non-reference"),
                  "runs": out}))
