#!/bin/bash
# GPU box: tools/pmc_paired.sh <label> -- SQ counters of the S2 job's kernels (bench.py --configs s2), one rocprofv3 --pmc pass
# per counter group (never combined with other tracing), per-kernel means by tools/pmc_summary.py.
set -u
label=${1:-s2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/pmc_$label
mkdir -p $out
declare -A groups
groups[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
groups[mem]="SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
cd /tmp
for g in ${PMC_GROUPS:-sq mem}; do
  timeout 300 rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d $out/$g -o p -- python3 $R/bench.py --no-cpu-baseline --configs s2 \
     --dist-variants 0 --graph-leg 0 --placement-experiment 0 --shard-bins 0 --steps 2 --warmup 1 --config-reps 2 "${@:2}" > $out/$g.log 2>&1 || echo "pass $g failed"
done
cd $R
python3 tools/pmc_summary.py $out k_bin_hist_s2 k_score_s2_bin k_score_s1_from_hist
