#!/bin/bash
# On the GPU box, from the repo root: tools/placement_pmc.sh <label>.  One rocprofv3 --pmc pass per counter group
# (TCC has four slots per pass; PMC passes carry --kernel-trace only), then the digest.
set -u
label=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/placement_$label
mkdir -p $out
declare -A groups
groups[a]="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
groups[b]="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum"
groups[c]="TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum"
groups[d]="TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
groups[e]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_LEVEL_VMEM"
groups[f]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_REQUEST_sum"
groups[g]="TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
groups[h]="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
cd /tmp
timeout 300 python3 $R/tools/placement_pmc.py > $out/plain.log 2>&1
for g in ${PLACEMENT_GROUPS:-a b c d e f g h}; do
  timeout 300 rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d $out/$g -o p -- python3 $R/tools/placement_pmc.py > $out/$g.log 2>&1 || echo "pass $g failed"
done
cd $R
python3 tools/placement_pmc_digest.py $out
