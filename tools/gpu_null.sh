#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/null
timeout 900 python -m pytest tests/test_hip_s3_null.py tests/test_hip_wide_models.py tests/test_hip_abi_calls.py -x -q -m gpu > gpurun_out/null/pytest.log 2>&1
echo "rc $?" >> gpurun_out/null/pytest.log
tail -4 gpurun_out/null/pytest.log
for w in 4 5; do
echo "--- EPG_NULL_WAVES=$w"
EPG_NULL_WAVES=$w timeout 600 python tools/null_ab.py 2>&1 | grep -v amdgpu.ids | grep "379 + 342" | tee -a gpurun_out/null/ab_waves.log
done
