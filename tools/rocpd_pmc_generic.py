"""Per-kernel averages of arbitrary PMC counters from a rocprofv3 rocpd database.

    python tools/rocpd_pmc_generic.py results.db [name-filter]
"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name").fetchall()
table = {}
for name, counter, n, avg in rows:
    if flt and flt not in name:
        continue
    table.setdefault(name, {})[counter] = (n, avg)
for name, cs in sorted(table.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
    print(name[:110])
    for c, (n, avg) in sorted(cs.items()):
        print(f"    {c:28s} launches {n:6d}  avg {avg:16.1f}")
