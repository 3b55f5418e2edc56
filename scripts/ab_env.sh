#!/bin/bash
# Same-box A/B of one environment switch: `scripts/ab_env.sh OUTDIR VAR A B [rounds]` runs bench.py (no CPU baseline, no extra routes)
# alternately with VAR=A and VAR=B, `rounds` times each (default 3), and prints img/s per run -- boxes of this pool differ by +-3 %,
# so step-level claims are made from interleaved pairs on one box.
OUT=$1; VAR=$2; A=$3; B=$4; R=${5:-3}
mkdir -p $OUT
for i in $(seq 1 $R); do
  for V in $A $B; do
    env $VAR=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-experiment > $OUT/ab_${VAR}_${V}_$i.json 2> $OUT/ab_${VAR}_${V}_$i.err || { tail -5 $OUT/ab_${VAR}_${V}_$i.err; exit 1; }
    python - <<PY
import json
d = json.load(open('$OUT/ab_${VAR}_${V}_$i.json'))
print('$VAR=$V run $i: %.1f img/s  %.2f ms/step  host %.1f ms  wgrad in-step %.4f alone %.4f' % (d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['roofline']['kernels']['res_wgrad']['ms_per_launch'], d['roofline']['kernels']['res_wgrad']['ms_per_launch_alone']))
PY
  done
done
