#!/usr/bin/env python3
"""Timeline of a `rocprofv3 --kernel-trace --output-format csv` run: where the device was idle.

    python tools/trace_timeline.py <..._kernel_trace.csv> [--pass-gap-us 800] [--gap-us 60]

Kernels separated by more than --pass-gap-us of idle device are taken as different passes (the host's work between two passes of the
integrator loop).  Per pass: span, device-busy time (union of the launches' intervals), the share of the span with 0 / 1 / 2 / 3 / 4+
launches in flight, the launches by kernel class, and the idle gaps longer than --gap-us with the kernel that ended before each."""
import argparse
import collections
import csv
import re


def cls(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:24]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--pass-gap-us", type=float, default=800.0)
    ap.add_argument("--gap-us", type=float, default=60.0)
    ap.add_argument("--max-gaps", type=int, default=12)
    a = ap.parse_args()
    rows = []
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])))
    rows.sort()
    passes, cur, end = [], [], None
    for s, e, k in rows:
        if end is not None and s - end > a.pass_gap_us*1e3:
            passes.append(cur); cur = []
        cur.append((s, e, k))
        end = e if end is None else max(end, e)
    if cur:
        passes.append(cur)
    prev_end = None
    for i, p in enumerate(passes):
        t0, t1 = p[0][0], max(e for _, e, _ in p)
        ev = sorted([(s, 1) for s, _, _ in p] + [(e, -1) for _, e, _ in p])
        depth, last, hist = 0, t0, collections.Counter()
        gaps, last_end_kernel = [], None
        ends = sorted((e, k) for _, e, k in p)
        for t, d in ev:
            hist[min(depth, 4)] += t - last
            if depth == 0 and t - last > a.gap_us*1e3 and d == 1:
                before = [k for e, k in ends if e <= last]
                gaps.append((last - t0, t - last, before[-1] if before else "-"))
            depth += d; last = t
        span = t1 - t0
        by = collections.Counter(); dur = collections.Counter()
        for s, e, k in p:
            by[k] += 1; dur[k] += e - s
        print("pass %d: starts %.2f ms after the previous one ended; span %.2f ms, %d launches; in flight 0/1/2/3/4+: %s"
              % (i, 0.0 if prev_end is None else (t0 - prev_end)/1e6, span/1e6, len(p),
                 " ".join("%.0f%%" % (100.0*hist[d]/max(span, 1)) for d in range(5))))
        print("   " + ", ".join("%s x%d %.2f ms" % (k, by[k], dur[k]/1e6) for k in sorted(by, key=lambda k: -dur[k])))
        for off, g, k in sorted(gaps, key=lambda x: -x[1])[:a.max_gaps]:
            print("   idle %.0f us at +%.2f ms (after %s)" % (g/1e3, off/1e6, k))
        prev_end = t1


if __name__ == "__main__":
    main()
