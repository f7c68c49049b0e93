"""Sums every counter of a rocprofv3 --pmc pass per kernel class.  usage: pmc_agg.py <counter_collection.csv> [<more.csv> ...]
Prints one JSON object: {kernel: {counter: total, ..., "launches": n}}."""
import csv
import json
import re
import sys
from collections import defaultdict


def main():
    sums = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for path in sys.argv[1:]:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                if not k.startswith("k_"):
                    continue
                k = re.sub(r"<.*", "", k)
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row.get("Dispatch_Id"))
    out = {}
    for k, c in sorted(sums.items()):
        out[k] = {n: round(v) for n, v in sorted(c.items())}
        out[k]["launches"] = len(launches[k])
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
