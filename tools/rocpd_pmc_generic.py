"""Per-kernel averages of arbitrary PMC counters from a rocprofv3 rocpd database.

    python tools/rocpd_pmc_generic.py results.db [name-filter]
"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)").fetchall()]
grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else None)  # (one kernel at several grids: the levels of a solve)
if grid:
    rows = cur.execute(f"select name || ' grid ' || {grid}, counter_name, count(*), avg(counter_value) from pmc_events group by name, {grid}, counter_name").fetchall()
else:
    rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name").fetchall()
table = {}
for name, counter, n, avg in rows:
    if flt and flt not in name:
        continue
    table.setdefault(name, {})[counter] = (n, avg)
for name, cs in sorted(table.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
    print(name[:60] + ' ...' + name[-24:] if len(name) > 90 else name)
    for c, (n, avg) in sorted(cs.items()):
        print(f"    {c:28s} launches {n:6d}  avg {avg:16.1f}")
