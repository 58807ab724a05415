"""HBM bytes per launch from two rocprofv3 PMC passes (rocpd sqlite output).

    python tools/rocpd_pmc.py fetch_results.db write_results.db > profiles/rXX_pmc_hbm_traffic_1M.txt

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B
(MI355X_MICROARCH.md, HBM section), so reads are doubled; the copy kernel of known size in the
same run (k_copy_d2: 16 B per site in, 16 B out) is printed first as the calibration.
"""
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), avg(counter_value) from pmc_events where counter_name = ? group by name", (counter,))
    return {r[0]: (r[1], r[2]) for r in rows}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
print("# corrected HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   [KiB counters; gfx950 FETCH_SIZE x2]")
print(f"{'kernel':96s} {'launches':>8s} {'FETCH_KiB':>12s} {'WRITE_KiB':>12s} {'HBM_MB':>10s}")
names = sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, (0, 0.0))[1]))
for k in names:
    n, f = fetch[k]
    w = write.get(k, (0, 0.0))[1]
    print(f"{k[:96]:96s} {n:8d} {f:12.1f} {w:12.1f} {(2 * f + w) * 1024 / 1e6:10.1f}")
