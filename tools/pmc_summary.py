#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv):

    python tools/pmc_summary.py gpurun_out/prof_v4/fetch gpurun_out/prof_v4/write gpurun_out/prof_v4/tcc > profiles/xyz.csv

Counter values of one dispatch are summed over the rows rocprofv3 writes for it (one per counter instance), then
averaged over the dispatches of a kernel with the same grid (the same kernel runs on A and on A', whose grids differ).
Only the product kernels of namespace fs:: (spmv_*, spmm_*, cbcsr_*, ata_*) are listed unless --all is given."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(fs::\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    everything = "--all" in sys.argv
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # (kernel, grid) -> counter -> dispatch -> value
    for d in args:
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k and (everything or re.match(r"fs::(spmv_|spmm_|cbcsr_|ata_|tiled_combine|strided_copy)", k)):
                    per[(k, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))][row["Counter_Name"]][row["Dispatch_Id"]] \
                        += float(row["Counter_Value"])
    counters = sorted({c for k in per for c in per[k]})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "workgroups", "dispatches"] + ["mean_" + c for c in counters])
    for k in sorted(per):
        n = max(len(v) for v in per[k].values())
        w.writerow([k[0], k[1], n] + ["%.3f" % (sum(per[k][c].values()) / len(per[k][c])) if per[k][c] else "" for c in counters])


if __name__ == "__main__":
    main()
