#!/bin/bash
# like ab_env.sh for any number of values: scripts/ab_env3.sh OUTDIR VAR rounds v1 v2 v3 ...
OUT=$1; VAR=$2; R=$3; shift 3
mkdir -p $OUT
for i in $(seq 1 $R); do
  for V in "$@"; do
    env $VAR=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-experiment > $OUT/ab_${VAR}_${V}_$i.json 2> $OUT/ab_${VAR}_${V}_$i.err || { tail -5 $OUT/ab_${VAR}_${V}_$i.err; exit 1; }
    python - <<PY
import json
d = json.load(open('$OUT/ab_${VAR}_${V}_$i.json'))
k = d['roofline']['kernels']
print('$VAR=$V run $i: %.1f img/s  %.2f ms/step  host %.1f / %.1f ms  fwd %.3f dgrad %.3f wgrad %.3f in-step' % (d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['host_issue_in_region_ms_per_step'], k['res_fwd']['ms_per_launch'], k['res_dgrad']['ms_per_launch'], k['res_wgrad']['ms_per_launch']))
PY
  done
done
