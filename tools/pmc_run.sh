#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <label> <kbench args...>
# Runs tools/kbench.py under rocprofv3 once per counter group (PMC passes must not be combined with tracing
# domains other than --kernel-trace) and leaves CSVs under gpurun_out/pmc_<label>/<group>/.
set -u
label=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/pmc_$label
mkdir -p $out
declare -A groups
groups[sq]="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"
groups[sq2]="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
groups[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
groups[tcp2]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
groups[ta]="TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"
groups[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
groups[tcc2]="TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_LEVEL_sum"
groups[lds]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
groups[fetch]="FETCH_SIZE"
groups[write]="WRITE_SIZE"
cd /tmp
# PMC_SCRIPT=tools/kbench_all.py and PMC_GROUPS="sq sq2 lds" select another driver script / a subset of the passes
script=${PMC_SCRIPT:-tools/kbench.py}
for g in ${PMC_GROUPS:-${!groups[@]}}; do
  timeout ${PMC_TIMEOUT:-120} rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d $out/$g -o p -- python3 $R/$script "$@" > $out/$g.log 2>&1 || echo "pass $g failed"
done
cd $R
python3 tools/pmc_summary.py $out
