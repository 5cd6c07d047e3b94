#!/bin/bash
# On the GPU box: tools/pmc_s3.sh <label> -- PMC passes over tools/kbench_all.py --what s3 (S3 expected + score, 1 M bins at
# N = 833), one rocprofv3 run per counter group, summary by tools/pmc_summary.py.
set -u
label=${1:-s3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/pmc_$label
mkdir -p $out
declare -A groups
groups[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
groups[mfma]="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"
groups[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
groups[clk]="GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES"
groups[fetch]="FETCH_SIZE"
groups[write]="WRITE_SIZE"
cd /tmp
for g in ${PMC_GROUPS:-sq mfma tcc fetch write}; do
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d $out/$g -o p -- python3 $R/tools/kbench_all.py --what s3 --s3-bins ${S3_BINS:-1000000} > $out/$g.log 2>&1 || echo "pass $g failed"
done
cd $R
python3 tools/pmc_summary.py $out k_s3
