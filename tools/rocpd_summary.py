"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a --stats style table.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db "comment line" > profiles/xxx.txt
"""
import sqlite3
import sys

db = sys.argv[1]
comment = sys.argv[2:] 
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
    "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
for c in comment:
    print("# " + c)
print(f"# source: {db} (kernel dispatch table); durations in ns; total kernel time {tot/1e6:.2f} ms")
print(f"{'Name':104s} {'Calls':>8s} {'TotalDurationNs':>16s} {'AverageNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Percentage':>10s}")
for r in rows:
    print(f"{r[0][:104]:104s} {r[1]:8d} {int(r[2]):16d} {r[3]:12.1f} {int(r[4]):10d} {int(r[5]):10d} {100*r[2]/tot:10.2f}")
