#!/bin/bash
# rocprofv3 --pmc passes over scripts/run_layer.py: pmc_layer.sh <tag> "<run_layer args>"
TAG=$1; ARGS=$2
export TMPDIR=/tmp
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    D=$GRAFT_REPO_ROOT/gpurun_out/pmcl_${TAG}_$(echo $C | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $C --kernel-trace -d $D -- python3 $GRAFT_REPO_ROOT/scripts/run_layer.py $ARGS > $D.log 2>&1 || tail -3 $D.log
    DB=$(find $D -name "*.db" | head -1)
    [ -n "$DB" ] && python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $DB "hgemm" 2>&1
done
