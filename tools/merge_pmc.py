#!/usr/bin/env python3
"""profiles/rNN_pmc_<config>.json for a round in which `rocprofv3 --pmc` does not survive the training bench:
the previous round's summary for the kernels that did not change, each entry marked with where it is from, plus the entries of
summaries collected on single operators this round (tools/collect_pmc_dcn.sh), which replace same-named ones.
usage: merge_pmc.py <previous round's json> <out json> <round tag> <operator json>..."""
import json
import sys

prev, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
base = json.load(open(prev))
res = {"note": base["note"] + f"  MERGED ({tag}): entries with \"source\" = the earlier round's file are that round's figures for kernels "
                              "that have not changed since; the others were collected this round on the operator alone.",
       "command": base["command"], "kernels": {}}
for k, v in base["kernels"].items():
    res["kernels"][k] = dict(v, source=prev.split("/")[-1])
for f in sys.argv[4:]:
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        res["kernels"][k] = dict(v, source=f"{tag}: {d['command']}")
json.dump(res, open(out, "w"), indent=1)
print(out, len(res["kernels"]), "kernels")
