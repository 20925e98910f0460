#!/usr/bin/env python3
"""L2-miss (fabric) bytes per launch of the dominant kernel from rocprofv3 PMC passes (tools/prof_pmc.sh
with PMC_MEM=1), for bench.py's roofline.traffic field.

Uses the L2's memory-side request counters by size, which need no unit correction:
  read  bytes = 32*TCC_EA0_RDREQ_32B + 64*TCC_EA0_RDREQ_64B + 128*TCC_EA0_RDREQ_128B
  write bytes = 64*TCC_EA0_WRREQ_64B + 32*(TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)
(cross-check: FETCH_SIZE [KiB] * 2, the gfx950 correction of MI355X_MICROARCH.md "HBM", agrees
with the read figure to <1 % on this kernel).

usage: make_traffic.py <pmc_dir> <config> <frames> <out.json>
The record carries sha256[:16] of the library the run loaded: bench.py reports `traffic` only for that very build.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

pmc, config, frames, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(pmc, "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            # one key per (kernel, grid): the same kernel also runs small single-plane launches (host-pointer ABI leg)
            acc[(row["Kernel_Name"], row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
name = max((k for k in acc if "remap_tiled" in k[0] or "remap_gather_kernel" in k[0]),
           key=lambda k: (lambda v: sum(v) / max(1, len(v)))(acc[k].get("TCC_EA0_RDREQ_sum", [0])), default=None)
if name is None:
    sys.exit("no gather kernel found in %s" % pmc)
m = {k: sum(v) / len(v) for k, v in acc[name].items()}
rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0))
res = {"config": config, "frames": frames, "grid": name[1],
       "kernel": (re.search(r"(\w+_kernel)", name[0]).group(1) if re.search(r"(\w+_kernel)", name[0]) else name[0]),
       "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
       "hbm_bytes_per_launch": int(rd + wr),
       "fetch_size_kib_x2_bytes": int(m.get("FETCH_SIZE", 0) * 1024 * 2),
       "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ_{,64B}_sum, mean over dispatches",
       "what": "L2-miss (fabric) bytes: requests the XCDs' L2s send to the memory side.  Infinity-Cache hits are INCLUDED "
               "(the counters sit in front of the MALL), so this is an upper bound of the HBM bytes"}
# all kernels of a step (config 3: the low-pass launches too).  Every counter comes from its own pass, and the passes do not
# see the same number of dispatches (r03: RDREQ 1 199 samples, WRREQ 975): each counter is averaged over ITS OWN samples,
# and a kernel's launches per step come from one reference pass (the RDREQ one) -- dividing every sum by the RDREQ sample
# count understated the writes by 19-25 % (VERDICT round 3, weak point 8).
ref = "TCC_EA0_RDREQ_sum"
steps = len(acc[name].get(ref, [])) or 1
tot_rd = tot_wr = 0.0
per_kernel = {}
for k, c in acc.items():
    if "fill_noise" in k[0] or "mapgen" in k[0]:
        continue
    per_step = len(c.get(ref, [])) / steps          # launches of this kernel per step
    g = lambda n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0) * per_step  # noqa: E731  mean per launch x launches per step
    krd = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
    kwr = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))
    short = re.search(r"(\w+_kernel)", k[0])
    key = "%s grid %s" % (short.group(1) if short else k[0][:40], k[1])
    per_kernel[key] = {"read_bytes_per_step": int(krd), "write_bytes_per_step": int(kwr),
                       "dispatches_per_step": round(per_step, 2)}
    tot_rd += krd
    tot_wr += kwr
res["step_bytes_all_kernels"] = int(tot_rd + tot_wr)
res["per_kernel"] = per_kernel
lib = os.environ.get("T360_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transform360_amd", "lib",
                                                 "libTransform360.so")
with open(lib, "rb") as f:
    res["library_sha16"] = hashlib.sha256(f.read()).hexdigest()[:16]
with open(out, "w") as f:
    json.dump(res, f, indent=1)
    f.write("\n")
print(json.dumps(res))
