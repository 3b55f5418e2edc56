#!/bin/bash
# one rocprofv3 --pmc pass over scripts/run_res_conv.py: pmc_pass.sh <tag> "<counters>"   (library through PCGAN_LIB)
TAG=$1; C=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $C --kernel-trace -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/run_res_conv.py 5 > $OUT.log 2>&1 || { tail -5 $OUT.log; }
DB=$(find $OUT -name "*.db" | head -1)
[ -n "$DB" ] && python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $DB "" > $OUT.txt 2>&1
grep -A6 "halo\|bsplit_conv" $OUT.txt | head -60
