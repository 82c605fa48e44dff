#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small markdown table for profiles/.
usage: rocpd_summary.py <results.db> [title]"""
import sqlite3
import sys

db = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else db
con = sqlite3.connect(db)
cur = con.cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
regs = {}
for name, vg, sg, lds, scr in cur.execute(
        "select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name"):
    regs[name] = (vg, sg, lds, scr)
print(f"# {title}\n")
print("rocprofv3 --kernel-trace --stats (durations in microseconds)\n")
print("| kernel | calls | total us | avg us | % | vgpr | sgpr | lds | scratch |")
print("|---|---|---|---|---|---|---|---|---|")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "").replace("ryujin_hip::", "")
    vg, sg, lds, scr = regs.get(name, ("", "", "", ""))
    print(f"| {short} | {calls} | {tot:.1f} | {avg:.1f} | {pct:.2f} | {vg} | {sg} | {lds} | {scr} |")
