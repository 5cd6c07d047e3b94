#!/bin/bash
# round end: the whole GPU suite, smoke(), then the bench + rocprofv3 + PMC passes of tools/profile_bench.sh
set -u
L=${1:-r03w}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$L
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$L/tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/$L/tests.log
tail -6 gpurun_out/$L/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_bench.sh $L 2>&1 | tail -30
