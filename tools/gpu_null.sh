#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/null
timeout 900 python -m pytest tests/test_hip_s3_null.py tests/test_hip_wide_models.py tests/test_hip_abi_calls.py -x -q -m gpu > gpurun_out/null/pytest.log 2>&1
echo "rc $?" >> gpurun_out/null/pytest.log
tail -15 gpurun_out/null/pytest.log
timeout 600 python tools/null_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/null/ab.log
