"""Mean of every collected PMC counter per kernel from a rocprofv3 counter_collection.csv.  usage: pmc_agg.py <csv>..."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            if not k.startswith("k_"): continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s %16.1f  (n=%d)" % (c, acc[k][c]/n[k][c], n[k][c]))
