#!/bin/bash
# On the GPU box, from the repo root: tools/profile_bench.sh <round-label>
# 1) plain bench line, 2) rocprofv3 --kernel-trace --stats of the same command, 3) separate PMC passes for
# FETCH_SIZE / WRITE_SIZE (never combined with other tracing domains).  Raw files land in gpurun_out/prof_<label>/;
# tools/profile_digest.py turns them into the summaries that get committed under profiles/.
set -u
label=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/prof_$label
mkdir -p $out
cd $R
timeout 600 python3 bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
# the other kernels of the path (S2, S3 expected/score, null shuffle) through tools/kbench_all.py
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_all -o p -- python3 $R/tools/kbench_all.py --what s2,s3,null,pair --bins 15000000 --s3-bins 1000000 --null-bins 1000000 > $out/stats_all.log 2>&1
cd $R
python3 tools/profile_digest.py $out $label
