"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/r1/traffic.json.

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: both counters are in KiB, and on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read (MI355X_MICROARCH.md "HBM"; cdna_hip_programming.md section 7), so the read
side is doubled as prescribed there.  The two counters come from separate passes (TCC slot limits).

usage: pmc_traffic.py <scene> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    sums, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            k = re.sub(r"<.*", "", k).replace("_dyn", "")   # kernel class: the dynamic-fetch variants count as their class
            sums[k] += float(row["Counter_Value"])
            n[k] += 1
    return {k: (sums[k]/n[k], n[k]) for k in sums}


def main():
    scene, fetch_csv, write_csv, out = sys.argv[1:5]
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    for k in sorted(set(fetch) & set(write)):
        f, nf = fetch[k]
        w, nw = write[k]
        data["%s/%s" % (scene, k)] = {
            "hbm_bytes_per_launch": round((2.0*f + w)*1024.0),
            "fetch_size_kib_per_launch": round(f, 1), "write_size_kib_per_launch": round(w, 1), "launches": [nf, nw],
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), read side doubled per MI355X_MICROARCH.md"}
        print("%-28s fetch %10.1f KiB  write %10.1f KiB  -> %.1f MB HBM per launch (%d launches)" % (k, f, w, (2*f + w)/1024.0, nf))
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
