#!/bin/bash
# bf16 step (BASELINE configs[2] per GPU): rocprofv3 kernel statistics, default streams and single-stream, grouped.  -> gpurun_out/$1/
set -e
TAG=${1:-r04_bf16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_under_profiler.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_bf16_kernel_stats.csv && rm -rf $OUT/stats
PCGAN_SIDE_STREAM=0 PCGAN_BRANCH_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ss -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_single_stream.json 2> $OUT/ss.err
cp $(find $OUT/ss -name "*kernel_stats.csv" | head -1) $OUT/single_stream_kernel_stats_bf16.csv && rm -rf $OUT/ss
python3 $GRAFT_REPO_ROOT/scripts/group_stats.py $OUT/single_stream_kernel_stats_bf16.csv 21 > $OUT/single_stream_groups_bf16.txt
cat $OUT/single_stream_groups_bf16.txt
cut -c1-300 $OUT/bench_single_stream.json
