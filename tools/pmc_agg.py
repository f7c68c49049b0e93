"""Sums every counter of a rocprofv3 --pmc pass per kernel class.  usage: pmc_agg.py [--variants] <counter_collection.csv> [<more.csv> ...]
(--variants keeps the template arguments: one entry per instantiation.)
Prints one JSON object: {kernel: {counter: total, ..., "launches": n}}."""
import csv
import json
import re
import sys
from collections import defaultdict


def main():
    sums = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    keep = "--variants" in sys.argv[1:]
    for path in [p for p in sys.argv[1:] if p != "--variants"]:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                if not k.startswith("k_"):
                    continue
                if not keep:
                    k = re.sub(r"<.*", "", k)
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row.get("Dispatch_Id"))
    out = {}
    for k, c in sorted(sums.items()):
        out[k] = {n: round(v) for n, v in sorted(c.items())}
        out[k]["launches"] = len(launches[k])
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            out[k]["lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"]/(64.0*c["SQ_ACTIVE_INST_VALU"]), 4)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
