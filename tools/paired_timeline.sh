#!/bin/bash
# GPU box, from the repo root: tools/paired_timeline.sh <label> -- rocprofv3 --kernel-trace of bench.py's paired job alone, then the
# timeline of its last repetition (which launches of the two streams really overlap, and what that does to their durations).
L=$1; shift
out=gpurun_out/$L
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o p -- python3 bench.py --no-cpu-baseline --configs paired --dist-variants 0 \
  --graph-leg 0 --placement-experiment 0 --shard-bins 0 --steps 3 --warmup 1 "$@" > $out/run.json 2> $out/run.err
python3 tools/trace_timeline.py $out/trace --last-ms ${LAST_MS:-60} --grep ${GREP:-k_bin_hist,k_null_hist,k_pair_fused,k_s1_combine} | cut -c1-120 > $out/timeline.txt
tail -60 $out/timeline.txt
rm -rf $out/trace
