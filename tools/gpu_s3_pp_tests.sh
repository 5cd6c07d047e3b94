#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ring
timeout 1500 python -m pytest tests/test_hip_s3_stress.py tests/test_hip_s3_n833.py -x -q -m gpu > gpurun_out/ring/pytest_s3.log 2>&1
echo "rc $?" >> gpurun_out/ring/pytest_s3.log
tail -6 gpurun_out/ring/pytest_s3.log
