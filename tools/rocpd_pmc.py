"""HBM bytes per launch from two rocprofv3 PMC passes (rocpd sqlite output).

    python tools/rocpd_pmc.py fetch_results.db write_results.db [--json profiles/rXX_pmc_hbm_traffic_1M.json] \
        > profiles/rXX_pmc_hbm_traffic_1M.txt

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B
(MI355X_MICROARCH.md, HBM section), so reads are doubled; a copy of known size in the same run
(`__amd_rocclr_copyBuffer` / k_extrapolate: 24 B per site in, 24 B out) serves as the calibration.
The optional JSON (kernel name -> corrected bytes per launch) is what bench.py reads for
`roofline.traffic`.
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), avg(counter_value) from pmc_events where counter_name = ? group by name", (counter,))
    return {r[0]: (r[1], r[2]) for r in rows}


argv = [a for a in sys.argv[1:]]
json_out = None
if "--json" in argv:
    k = argv.index("--json")
    json_out = argv[k + 1]
    del argv[k:k + 2]
fetch = per_kernel(argv[0], "FETCH_SIZE")
write = per_kernel(argv[1], "WRITE_SIZE")
for c in argv[2:]:
    print("# " + c)
print("# corrected HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   [KiB counters; gfx950 FETCH_SIZE x2]")
print(f"{'kernel':96s} {'launches':>8s} {'FETCH_KiB':>12s} {'WRITE_KiB':>12s} {'HBM_MB':>10s}")
names = sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, (0, 0.0))[1]))
table = {}
for k in names:
    n, f = fetch[k]
    w = write.get(k, (0, 0.0))[1]
    table[k] = (2 * f + w) * 1024
    print(f"{k[:96]:96s} {n:8d} {f:12.1f} {w:12.1f} {(2 * f + w) * 1024 / 1e6:10.1f}")
if json_out:
    with open(json_out, "w") as fh:
        json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two passes), (2*FETCH+WRITE)*1024 bytes",
                       comment=argv[2:], hbm_bytes_per_launch=table), fh, indent=1)
