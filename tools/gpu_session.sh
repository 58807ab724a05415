#!/bin/bash
# final-commit evidence: parity lines of the other workloads, a 40k-step soak at 1M
mkdir -p gpurun_out
: > gpurun_out/BENCH_r04D_parity_lines.jsonl
for W in 5k 60k 250k strip500k strip500k_ff; do
  timeout 900 python bench.py --workload $W --steps 20 --warmup 5 2>gpurun_out/r04D_$W.err >> gpurun_out/BENCH_r04D_parity_lines.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/BENCH_r04D_parity_lines.jsonl'):
    d=json.loads(l)
    par=[d.get('parity_vs_oracle')]+[v.get('parity_vs_oracle') for v in d.values() if isinstance(v,dict) and 'parity_vs_oracle' in v]
    worst=max(max(p[k] for k in ('dt','abs_sq_psi','mu_zero_mean','J_s','J_n')) for p in par if p)
    print(d['config']['workload'][:40], d['value'], d.get('setup_s',{}).get('total'), 'sustained', (d.get('sustained') or {}).get('value'), 'worst parity', worst, all(p['ok'] for p in par if p))
PY
timeout 1500 python tools/soak.py 1M 40000 > gpurun_out/SOAK_r04D.jsonl 2> gpurun_out/soak_r04D.err
tail -2 gpurun_out/SOAK_r04D.jsonl | cut -c1-600
