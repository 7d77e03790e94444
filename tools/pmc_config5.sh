cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C -d /tmp/p_$T -o c5 -- python3 tools/bench_config5.py 4096 > /tmp/p_$T.log 2>&1
  DB=$(find /tmp/p_$T -name '*.db' | head -1)
  if [ -n "$DB" ]; then python3 tools/rocpd_pmc.py "$DB" 2>&1 | grep -i "nuts_kernel" | cut -c60-200; else echo "no db for $C"; tail -3 /tmp/p_$T.log; fi
done
