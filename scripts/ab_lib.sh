#!/bin/bash
# Same-box A/B of two builds of the library (PCGAN_LIB=path): `scripts/ab_lib.sh OTHER.so [rounds]` runs bench.py alternately with the
# in-tree library and OTHER.so and prints img/s per run; then the kernel dashboard's hgemm rows for both.
OTHER=$1; R=${2:-3}
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for V in "" "$OTHER"; do
    env PCGAN_LIB=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-experiment > gpurun_out/_abl.json 2> gpurun_out/_abl.err || { tail -5 gpurun_out/_abl.err; exit 1; }
    python - "$V" <<'PY'
import json, sys
d = json.load(open('gpurun_out/_abl.json'))
print('%-40s %.1f img/s  %.2f ms/step' % (sys.argv[1] or 'in-tree', d['value'], d['ms_per_step']))
PY
  done
done
for V in "" "$OTHER"; do
  echo "== dashboard, ${V:-in-tree}"
  env PCGAN_LIB=$V python scripts/bench_kernels.py 2>/dev/null | grep -E "layer|G.res|G.down|G.up|D.c[1-3]|E.l|IP.c[2-5]|sum of"
done
