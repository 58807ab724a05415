#!/bin/bash
# Same-box A/B of bench.py variants: bash tools/gpu_ab.sh TAG "args A" "args B" ...
# One JSON line per variant in gpurun_out/AB_<TAG>.jsonl (value, ms/step, iterations).
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/AB_${TAG}.jsonl
for V in "$@"; do
  # "ENV=1 OTHER=2 @ --args": environment assignments before the @
  E=""; A="$V"
  case "$V" in *@*) E="${V%%@*}"; A="${V#*@}";; esac
  timeout 900 env $E python bench.py --no-cpu-baseline --vortex-window off $A > $OUT/ab_tmp.json 2> $OUT/ab_${TAG}_last.err || { echo "FAILED: $V"; tail -5 $OUT/ab_${TAG}_last.err; }
  python - "$V" <<'PY' >> $OUT/AB_${TAG}.jsonl
import json,sys
try:
    d=json.load(open('gpurun_out/ab_tmp.json'))
    print(json.dumps(dict(variant=sys.argv[1], value=d['value'], ms=d['ms_per_step'], its=d['pcg']['mean_iterations'], pred=d['pcg'].get('batch_prediction'), k1=d['roofline']['frac'], axp_ms=(d.get('roofline_pcg') or {}).get('avg_launch_ms'), agg=d['step_aggregate']['frac_of_hbm_peak'])))
except Exception as e:
    print(json.dumps(dict(variant=sys.argv[1], error=str(e))))
PY
  tail -1 $OUT/AB_${TAG}.jsonl
done
