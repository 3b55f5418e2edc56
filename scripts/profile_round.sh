#!/bin/bash
# The round's measurement set on the GPU box (run from the repo root through gpurun): bench line, rocprofv3 kernel statistics of
# the same command, and separate --pmc passes over the dominant convolution.  Everything lands under gpurun_out/$1/.
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
cat $OUT/bench_n1.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > $OUT/bench_under_profiler.json 2> $OUT/stats.err
cat $OUT/bench_under_profiler.json
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace -d $D -- python3 $GRAFT_REPO_ROOT/scripts/run_res_conv.py 5 > $D.log 2>&1 || { tail -5 $D.log; }
    DB=$(find $D -name "*.db" | head -1)
    [ -n "$DB" ] && python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $DB "" > $D.txt 2>&1 || true
done
cat $OUT/pmc_*.txt | grep -v "^$" | head -120
