#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/BENCH_r03n_parity_lines.jsonl
for W in 1M 250k strip500k strip500k_ff 5k; do
  timeout 1500 python bench.py --workload $W --steps 20 --warmup 5 > $OUT/tmp_line.json 2> $OUT/r03n_$W.err
  echo "$W rc=$?"; cat $OUT/tmp_line.json >> $OUT/BENCH_r03n_parity_lines.jsonl
  python - <<'PY'
import json
d=json.load(open('gpurun_out/tmp_line.json'))
print(d['config']['sites'], d['value'], 'cpu', d['cpu_baseline']['value'], 'x', d['speedup_vs_cpu_baseline'])
print('  parity', {k: (f"{v:.1e}" if isinstance(v,float) else v) for k,v in d['parity_vs_oracle'].items() if k not in ('source','scales')})
vw=d.get('vortex_window') or {}
print('  vortex', vw.get('value'), vw.get('state_reached'), {k: (f"{v:.1e}" if isinstance(v,float) else v) for k,v in (vw.get('parity_vs_oracle') or {}).items() if k not in ('source','scales')})
PY
done
