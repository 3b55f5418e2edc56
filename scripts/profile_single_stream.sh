#!/bin/bash
# single-stream kernel statistics (side / branch streams off: per-kernel durations are those of a kernel running alone) of the
# bench step, fp32 and bf16; results under gpurun_out/$1/
TAG=${1:-r02ss}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PCGAN_SIDE_STREAM=0 PCGAN_BRANCH_STREAMS=0
cd /tmp
for DT in fp32 bf16; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$DT -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_$DT.json 2> $OUT/$DT.err
    F=$(find $OUT/$DT -name "*kernel_stats.csv" | head -1)
    cp $F $OUT/kernel_stats_$DT.csv
    tail -1 $OUT/bench_$DT.json | cut -c1-200
done
