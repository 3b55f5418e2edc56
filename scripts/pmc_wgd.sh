cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TA_[A-Z_a-z]*\|TCP_[A-Z_a-z]*\|SQ_INST[A-Z_a-z]*\|SQ_WAIT[A-Z_a-z]*\|SQ_BUSY[A-Z_a-z]*\|SQ_ACTIVE_INST[A-Z_a-z]*" | sort -u | tr '\n' ' ' | cut -c1-3000
echo
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$N -- python3 $GRAFT_REPO_ROOT/scripts/time_wgd.py > /tmp/pmc_$N.log 2>&1 || tail -3 /tmp/pmc_$N.log
  DB=$(find /tmp/pmc_$N -name "*.db" | head -1)
  [ -n "$DB" ] && python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $DB "" 2>&1 | grep -A8 "wgd_main\|hsplit_wgrad" | head -40
done
