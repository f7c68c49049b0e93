"""Aggregates a rocprofv3 --pmc pass over SQ counters into profiles/sq_counters.json: where the waves of each kernel class
spend their cycles.  MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_WAIT_ANY (wave parked on s_waitcnt / barrier) +
SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing) ~= SQ_WAVE_CYCLES, all in quad-cycles.

usage: pmc_sq.py <scene> <counter_collection.csv> <out.json>"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def main():
    scene, path, out = sys.argv[1:4]
    sums = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            if not k.startswith("k_"):
                continue
            k = re.sub(r"<.*", "", k).replace("_dyn", "")
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row.get("Dispatch_Id"))
    data = json.load(open(out)) if os.path.exists(out) else {}
    for k, c in sorted(sums.items()):
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        entry = {name: round(v) for name, v in sorted(c.items())}
        entry["launches"] = len(launches[k])
        if wave > 0:
            for name, label in (("SQ_WAIT_ANY", "frac_waiting"), ("SQ_WAIT_INST_ANY", "frac_issue_stalled"),
                                ("SQ_ACTIVE_INST_ANY", "frac_issuing"), ("SQ_ACTIVE_INST_VALU", "frac_issuing_valu")):
                if name in c:
                    entry[label] = round(c[name]/wave, 4)
        data["%s/%s" % (scene, k)] = entry
        print("%-24s %s" % (k, {n: v for n, v in entry.items() if n.startswith("frac")}))
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
