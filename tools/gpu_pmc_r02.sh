#!/bin/bash
# PMC passes on the step kernel (one counter set per pass, --kernel-trace only: see the guide's rocprofv3 section)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r02k}
mkdir -p $O
P1="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TA_BUSY_avr"
P3="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum"
for m in ${2:-3 2}; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P -d $O/m${m}_p$i -o r -- python bench.py --steps 30 --warmup 60 --no-cpu --step-mode $m --groups 1 > $O/m${m}_p$i.json 2> $O/m${m}_p$i.err
    echo "== mode $m pass $i rc $?"
    python tools/rocprof_summary.py $(find $O/m${m}_p$i -name "r_results.db") 2>&1 | grep "mrx_k_cim_step\|mrx_k_cim_sched"
  done
done
