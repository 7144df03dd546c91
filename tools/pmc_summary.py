"""Summarise rocprofv3 --pmc CSVs: per-kernel average counter value per dispatch.
usage: pmc_summary.py <counter_collection.csv> [...]  → JSON {kernel: {counter: avg, dispatches: n}}"""
import collections
import csv
import json
import sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name")
        out[name.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
def active_mean(v):
    # gated-off launches of the LM state machine return at once and move (almost) nothing: average the launches that did work
    thr = 0.01 * max(v) if v else 0.0
    act = [x for x in v if x > thr] or v
    return sum(act) / len(act), len(act)


res = {}
for k, d in out.items():
    res[k] = {c: active_mean(v)[0] for c, v in d.items()}
    res[k]["dispatches"] = max(len(v) for v in d.values())
    res[k]["active_dispatches"] = max(active_mean(v)[1] for v in d.values())
print(json.dumps(res, indent=1, sort_keys=True))
