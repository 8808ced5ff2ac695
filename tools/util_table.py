#!/usr/bin/env python3
"""Per-kernel utilisation table of one pretrain step from rocprofv3 passes over bench.py:
   average duration (kernel trace), HBM GB/s (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction), MFMA pipe utilisation
   = SQ_INSTS_VALU_MFMA (wave-level v_mfma_f32_16x16x32_bf16 count) x 16 cycles / (1024 SIMDs x kernel cycles), with the
   kernel's cycles from GRBM_GUI_ACTIVE / 8 XCDs, and VALU issue share = SQ_INSTS_VALU x 4 / (1024 x cycles).
   python tools/util_table.py <pmc_dir with passN/ counter csvs> <pmc_hbm json> <kernel_stats.csv> <steps>"""
import csv, glob, json, re, sys
from collections import defaultdict


def key(name):
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", name)).replace("mae::", "")
    k = re.sub(r"^_ZN3mae\d+", "", k)
    return re.sub(r"(ILi|IDF|IfL|EvP|<).*", "", k)


cnt = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{sys.argv[1]}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = cnt[key(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
hbm = json.load(open(sys.argv[2]))
steps = int(sys.argv[4])
dur = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[3])):
    d = dur[key(r["Name"])]
    d[0] += float(r["TotalDurationNs"]); d[1] += int(r["Calls"])
rows = []
for k, (tot, calls) in dur.items():
    c = cnt.get(k, {})
    per = lambda n: (c[n][0] / c[n][1]) if n in c and c[n][1] else None
    cyc = per("GRBM_GUI_ACTIVE")
    cyc = cyc / 8 if cyc else None
    mf, va = per("SQ_INSTS_MFMA"), per("SQ_INSTS_VALU")
    hb = next((v["hbm_bytes_per_launch"] for kk, v in hbm.items() if key(kk) == k), None)
    avg_us = tot / calls / 1e3
    rows.append((tot / steps / 1e6, k, calls / steps, avg_us, hb / (avg_us * 1e3) if hb else None,
                 mf * 16 / (1024 * cyc) if mf is not None and cyc else None, va * 4 / (1024 * cyc) if va is not None and cyc else None,
                 cyc / avg_us / 1e3 if cyc else None))
print(f"{'kernel':34s} {'ms/step':>8s} {'launches':>8s} {'avg us':>8s} {'HBM GB/s':>9s} {'MFMA util':>9s} {'VALU issue':>10s} {'GHz':>5s}")
for ms, k, n, us, gbs, mu, vu, ghz in sorted(rows, reverse=True)[:24]:
    f = lambda v, fmt: format(v, fmt) if v is not None else "-"
    print(f"{k[:34]:34s} {ms:8.3f} {n:8.1f} {us:8.1f} {f(gbs, '9.0f'):>9s} {f(mu, '9.2f'):>9s} {f(vu, '10.2f'):>10s} {f(ghz, '5.2f'):>5s}")
