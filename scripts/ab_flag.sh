#!/bin/bash
# Same-box A/B of one bench.py FLAG: `scripts/ab_flag.sh OUTDIR TAG "flags A" "flags B" [rounds]`
OUT=$1; TAG=$2; A=$3; B=$4; R=${5:-3}
mkdir -p $OUT
for i in $(seq 1 $R); do
  for V in A B; do
    F=$A; [ $V = B ] && F=$B
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-experiment $F > $OUT/ab_${TAG}_${V}_$i.json 2> $OUT/ab_${TAG}_${V}_$i.err || { tail -5 $OUT/ab_${TAG}_${V}_$i.err; exit 1; }
    python - <<PY
import json
d = json.load(open('$OUT/ab_${TAG}_${V}_$i.json'))
k = d['roofline']['kernels']
print('$TAG $V [$F] run $i: %.1f img/s  %.2f ms/step  host %.1f / %.1f ms  fwd %.3f dgrad %.3f wgrad %.3f in-step' % (d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['host_issue_in_region_ms_per_step'], k['res_fwd']['ms_per_launch'], k['res_dgrad']['ms_per_launch'], k['res_wgrad']['ms_per_launch']))
PY
  done
done
