#!/bin/bash
# Socket power and shader clock while bench.py runs (rocm-smi sampled every 0.5 s beside a 150-step run): is the step power-limited?
# usage: bash scripts/power_sample.sh [bench.py flags, e.g. --dtype bf16]   -> gpurun_out/power_<tag>.txt
TAG=$(echo "fp32 $*" | tr -c 'a-zA-Z0-9\n' '_' | sed 's/_*$//')
OUT=gpurun_out/power_$TAG.txt
mkdir -p gpurun_out
( for i in $(seq 1 16); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk clock level" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT &
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-experiment "$@" 2>&1 >/dev/null | grep "timed region done" | tee -a $OUT.run
wait
cat $OUT.run >> $OUT; rm -f $OUT.run
cat $OUT
