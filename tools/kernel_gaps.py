"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: per (previous kernel -> next kernel) pair the mean gap
(start of the next minus end of the previous) and the kernels' mean durations, over the last N dispatches.
    python tools/kernel_gaps.py <..._kernel_trace.csv> [N]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+)(<[^(]*>)?\(", name)
    return (m.group(1) if m else name)[:40]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Kind") == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = rows[-n:]
    gaps, durs = defaultdict(list), defaultdict(list)
    prev = None
    for r in rows:
        s, e, k = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
        durs[k].append((e - s) / 1e3)
        if prev is not None:
            gaps[(prev[1], k)].append((s - prev[0]) / 1e3)
        prev = (e, k)
    print("kernel, calls, mean_us")
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        print("%s, %d, %.1f" % (k, len(v), sum(v) / len(v)))
    print("previous -> next, count, mean_gap_us, min_gap_us, max_gap_us")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        print("%s -> %s, %d, %.1f, %.1f, %.1f" % (a, b, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
