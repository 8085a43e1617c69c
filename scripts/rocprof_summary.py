#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db as a per-kernel table
(calls, avg/min/max ns, total ms, %), like `--stats` would print.
usage: rocprof_summary.py results.db [label] > profiles/xxx.md"""
import sqlite3, sys

db = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else db
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute(
    "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
    "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(scratch_size) "
    "from kernels group by name order by 6 desc"))
tot = sum(r[5] for r in rows) or 1
print(f"# rocprofv3 --kernel-trace --stats summary: {label}\n")
print("| kernel | calls | avg ns | min ns | max ns | total ms | % | grid | wg | lds B | vgpr | scratch |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for r in rows:
    n = r[0].replace("l2z::(anonymous namespace)::", "").replace("void ", "")
    print(f"| `{n[:70]}` | {r[1]} | {r[2]:.0f} | {r[3]:.0f} | {r[4]:.0f} | {r[5]/1e6:.2f} | "
          f"{100*r[5]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
