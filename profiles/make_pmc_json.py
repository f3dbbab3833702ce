#!/usr/bin/env python3
"""profiles/pmc.json: what bench.py merges into `roofline` (traffic, MFMA-busy fraction, executed FLOPs, the PMC pass's
clock) - collected from the pmc_summary.json files profiles/summarize_pmc.py wrote for the profile directories of one or
more rounds (a later round's directory replaces an earlier one's for the same workload; `_round` says which one it is).
usage: python profiles/make_pmc_json.py r05          (directories profiles/r05_<key>/ -> keys b256, b1024, cfg4_rf9,
                                                       eval_b4096, ...)
       python profiles/make_pmc_json.py r04 r05      (r04's, then r05's on top)"""
import glob
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
rounds = sys.argv[1:] or ["r06"]
out = {"_source": "rocprofv3 --pmc passes of profiles/prof_recipe.sh (separate passes; FETCH_SIZE x2 gfx950 correction, SQ quad-cycle "
                  "counters x4), summarised per kernel name by profiles/summarize_pmc.py; key = workload of the bench line "
                  "(b<windows> for BASELINE configs[1]'s shape, eval_b<windows> for one clip call of --mode eval, else bench.py --workload)",
       "workloads": {}}
for rnd in rounds:
    for d in sorted(glob.glob(os.path.join(here, rnd + "_*"))):
        f = os.path.join(d, "pmc_summary.json")
        if not os.path.exists(f):
            continue
        key = os.path.basename(d)[len(rnd) + 1:]
        s = json.load(open(f))
        s["_table"] = "profiles/%s/pmc_table.txt" % os.path.basename(d)
        s["_round"] = rnd
        out["workloads"][key] = s
json.dump(out, open(os.path.join(here, "pmc.json"), "w"), indent=1)
print("profiles/pmc.json:", ", ".join("%s (%s)" % (k, v["_round"]) for k, v in out["workloads"].items()))
