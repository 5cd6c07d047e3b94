#!/bin/bash
# GPU box, from the repo root: tools/prof_stats.sh <label> <python script and its arguments...>
# rocprofv3 --kernel-trace --stats (csv) of one command; prints the script's last lines and the top of the kernel statistics.
L=$1; shift
out=gpurun_out/$L
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 "$@" > $out/run.log 2>&1
tail -${PROF_TAIL:-12} $out/run.log
f=$(find $out/stats -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -${PROF_TOP:-16} "$f" | cut -c1-220; cp "$f" $out/kernel_stats.csv; else echo "no kernel_stats.csv under $out/stats"; fi
rm -rf $out/stats
