#!/bin/bash
# Clock and socket power of single kernels running for seconds (scripts/power_kernel_loop.py beside rocm-smi every 0.5 s).  -> gpurun_out/power_kernels.txt
OUT=gpurun_out/power_kernels.txt
mkdir -p gpurun_out; : > $OUT
for K in fwd dgrad wgrad wgrad_pertap hgemm in_bwd; do
  ( sleep 3.5; for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk clock level" | sed 's/GPU\[0\]\t\t: //; s/Current Socket Graphics Package Power (W)/W/; s/sclk clock level: [0-9S]*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/_pk.txt &
  python scripts/power_kernel_loop.py $K 5 2>/dev/null | tee -a $OUT
  wait
  sed 's/^/    /' gpurun_out/_pk.txt >> $OUT
done
cat $OUT
