#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <title> <db> [<db> ...]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024."""
import sqlite3
import sys
from collections import defaultdict

title, dbs = sys.argv[1], sys.argv[2:]
vals = defaultdict(dict)
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "group by kernel_name, counter_name"):
        short = k.split("(")[0].replace("void ", "").replace("ryujin_hip::", "")
        vals[short][c] = v
        vals[short]["dispatches"] = n
counters = sorted({c for d in vals.values() for c in d if c != "dispatches"})
print(f"# {title}\n")
print("mean per dispatch; hbm_MB = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / 1e6 (gfx950 correction)\n")
print("| kernel | n | " + " | ".join(counters) + " | hbm_MB |")
print("|---|---|" + "---|" * (len(counters) + 1))
for k, d in sorted(vals.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    if not k.startswith("k_"):
        continue
    hbm = ""
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        hbm = f"{(2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / 1e6:.1f}"
    print(f"| {k} | {d['dispatches']} | " + " | ".join(f"{d.get(c, float('nan')):.4g}" for c in counters) + f" | {hbm} |")
