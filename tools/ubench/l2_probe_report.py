#!/usr/bin/env python3
"""tools/ubench/l2_probe_report.py DIR -- per-dispatch counters of l2_probe.bin from a rocprofv3 --pmc output dir."""
import csv
import glob
import os
import sys
from collections import OrderedDict

d = OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = int(r["Dispatch_Id"])
            e = d.setdefault(k, {"name": r["Kernel_Name"].split("(")[0], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(d):
    e = d[k]
    cs = " ".join("%s=%.0f" % (c.replace("TCC_", "").replace("_sum", ""), v) for c, v in e.items() if c not in ("name", "us"))
    print("%3d %-28s %8.1f us  %s" % (k, e["name"], e["us"], cs))
