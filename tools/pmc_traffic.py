#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py to HBM bytes per launch for each kernel family.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads -> x2;
WRITE_SIZE is exact for 16-byte stores.  Both counters are in KiB.   python tools/pmc_traffic.py DIR_FETCH DIR_WRITE OUT.json"""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, name):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"<.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("mae::", "")
        k = re.sub(r"^_ZN3mae\d+", "", k)
        k = re.sub(r"(ILi|IDF|IfL|EvP).*", "", k)
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    n = max(fetch[k][1], write[k][1])
    rd = 2 * 1024 * fetch[k][0] / max(1, fetch[k][1])
    wr = 1024 * write[k][0] / max(1, write[k][1])
    out[k] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
# whole-step sum: every launch of every kernel in the profiled run / the number of steps it ran (argv[4], default 3 = 1 warm-up + 2)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values()) / steps
out["_step"] = {"hbm_bytes_per_step": step_bytes, "steps_in_profile": steps,
                "workload": sys.argv[5] if len(sys.argv) > 5 else "bench.py default (ViT-S/8 96px MAE, batch 2000)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"HBM bytes per step (all kernels): {step_bytes / 1e9:.1f} GB")
out.pop("_step")
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:32s} launches {v['launches']:5d}  read {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB")
