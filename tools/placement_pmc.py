#!/usr/bin/env python3
"""post-processing for tools/placement_probe2.py under rocprofv3 --pmc: per group of consecutive dispatches of one kernel, the mean
duration and the mean of every counter.   python tools/placement_pmc.py <dir with *counter_collection.csv>"""
import csv
import glob
import sys
from collections import OrderedDict

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][:40], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
    d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
groups = []
for i, d in disp.items():
    if groups and groups[-1]["k"] == d["k"]:
        groups[-1]["d"].append(d)
    else:
        groups.append({"k": d["k"], "d": [d]})
for g in groups:
    if len(g["d"]) < 10 or not ("expand" in g["k"] or "reduce" in g["k"]):
        continue
    n = len(g["d"])
    names = sorted(g["d"][0]["c"])
    print(g["k"], n, "us %.1f" % (sum(x["t"] for x in g["d"]) / n), " ".join("%s %.4g" % (c, sum(x["c"][c] for x in g["d"]) / n) for c in names))
