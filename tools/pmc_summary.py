#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes written by tools/prof_pmc.sh: per kernel name, the mean of
every counter over its dispatches (counter_collection.csv: one row per dispatch per counter)."""
import csv
import glob
import re
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "t360" not in name:
                continue
            m = re.search(r"(\w+_kernel)", name)
            short = m.group(1) if m else name.split("(")[0].split("::")[-1]
            key = (short, row.get("Grid_Size", ""), row.get("VGPR_Count", row.get("Arch_VGPR_Count", "")),
                   row.get("LDS_Block_Size", ""), row.get("Scratch_Size", ""))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in sorted(acc):
    print("kernel=%s grid=%s vgpr=%s lds=%s scratch=%s" % key)
    for cname in sorted(acc[key]):
        v = acc[key][cname]
        print("    %-24s mean %16.1f  (n=%d)" % (cname, sum(v) / len(v), len(v)))
