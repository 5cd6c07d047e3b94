#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pair
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_chr1_full.py tests/test_pairwise_step4.py tests/test_hip_pipeline.py tests/test_hip_wide_models.py -x -q -m gpu -k "pair or Pair or chr1 or wide" > gpurun_out/pair/pytest.log 2>&1
echo "rc $?" >> gpurun_out/pair/pytest.log
tail -5 gpurun_out/pair/pytest.log
for w in 6 8 10 12; do
  echo "--- EPG_PAIR_WAVES=$w"
  EPG_PAIR_WAVES=$w timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --configs paired --placement-experiment 0 --config-reps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['paired']
print(d['job_ms'], d['phases_ms'])"
done
