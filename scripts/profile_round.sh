#!/bin/bash
# The round's measurement set on the GPU box (run from the repo root through gpurun): bench line, rocprofv3 kernel statistics of
# the same command (default streams) and single-stream, kernel dashboard, and separate --pmc passes over the three residual-convolution
# kernels.  Everything lands under gpurun_out/$1/.
set -e
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err || tail -5 $OUT/bench_bf16.err
cut -c1-300 $OUT/bench_bf16.json
python scripts/bench_kernels.py > $OUT/kernel_dashboard_fp32.txt 2>&1 || true
tail -3 $OUT/kernel_dashboard_fp32.txt
python scripts/host_issue.py > $OUT/host_issue.txt 2>&1 || true
grep composite $OUT/host_issue.txt || true
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_under_profiler.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv && rm -rf $OUT/stats
PCGAN_SIDE_STREAM=0 PCGAN_BRANCH_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ss -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_single_stream.json 2> $OUT/ss.err
cp $(find $OUT/ss -name "*kernel_stats.csv" | head -1) $OUT/single_stream_kernel_stats_fp32.csv && rm -rf $OUT/ss
python3 $GRAFT_REPO_ROOT/scripts/group_stats.py $OUT/single_stream_kernel_stats_fp32.csv 21 > $OUT/single_stream_groups.txt
cat $OUT/single_stream_groups.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    D=$OUT/pmcdir_$N
    rocprofv3 --pmc $C --kernel-trace -d $D -- python3 $GRAFT_REPO_ROOT/scripts/run_res_conv.py 5 > $OUT/pmc_$N.log 2>&1 || { tail -5 $OUT/pmc_$N.log; }
    DB=$(find $D -name "*.db" | head -1)
    [ -n "$DB" ] && python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $DB "" > $OUT/pmc_$N.txt 2>&1 || true
    rm -rf $D
done
( for N in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE_SQ_WAVE; do echo "== rocprofv3 --pmc $N --kernel-trace -- python3 scripts/run_res_conv.py 5"; cat $OUT/pmc_$N.txt; done ) > $OUT/counters_residual_convs.txt 2>/dev/null || true
python3 $GRAFT_REPO_ROOT/scripts/make_traffic_json.py $OUT $OUT/residual_kernel_traffic.json f16x2 r04 > /dev/null || true
cut -c1-600 $OUT/residual_kernel_traffic.json
# the bench line LAST: it reads the traffic / matrix-pipe figures of THIS run's counter passes (profiles/<tag>_residual_kernel_traffic.json)
cp $OUT/residual_kernel_traffic.json $GRAFT_REPO_ROOT/profiles/r04_residual_kernel_traffic.json || true
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
cut -c1-400 $OUT/bench_n1.json
