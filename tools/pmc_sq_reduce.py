#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters collected by tools/pmc_sq.sh.   python tools/pmc_sq_reduce.py gpurun_out/pmc_sq [kernel-name substring]"""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{sys.argv[1]}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("mae::", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        k = k[:70] + f" grid={r.get('Grid_Size', '?')}"
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k)
    print("   " + "  ".join(f"{n}={v[0] / v[1]:.3g}" for n, v in sorted(cs.items())))
