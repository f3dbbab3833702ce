#!/usr/bin/env python3
"""profiles/pmc.json: what bench.py merges into `roofline` (traffic, MFMA-busy fraction, executed FLOPs, the PMC pass's
clock) - collected from the pmc_summary.json files profiles/summarize_pmc.py wrote for a round's profile directories.
usage: python profiles/make_pmc_json.py r04     (directories profiles/r04_<key>/ -> keys b256, b1024, cfg4_rf9, ...)"""
import glob
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {"_source": "rocprofv3 --pmc passes of profiles/prof_recipe.sh (separate passes; FETCH_SIZE x2 gfx950 correction, SQ quad-cycle "
                  "counters x4), summarised per kernel name by profiles/summarize_pmc.py; key = workload of the bench line "
                  "(b<windows> for BASELINE configs[1]'s shape, else bench.py --workload)", "workloads": {}}
for d in sorted(glob.glob(os.path.join(here, rnd + "_*"))):
    f = os.path.join(d, "pmc_summary.json")
    if not os.path.exists(f):
        continue
    key = os.path.basename(d)[len(rnd) + 1:]
    s = json.load(open(f))
    s["_table"] = "profiles/%s/pmc_table.txt" % os.path.basename(d)
    out["workloads"][key] = s
json.dump(out, open(os.path.join(here, "pmc.json"), "w"), indent=1)
print("profiles/pmc.json:", ", ".join(out["workloads"]))
