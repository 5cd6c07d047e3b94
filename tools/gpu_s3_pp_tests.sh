#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ring
EPG_S3_SYRK=pp timeout 1500 python -m pytest tests/test_hip_s3_stress.py tests/test_hip_s3_n833.py tests/test_hip_s3_null.py tests/test_hip_parity.py tests/test_hip_abi_calls.py tests/test_hip_pipeline.py -x -q -m gpu -k "s3 or S3 or sal or driver or cli" > gpurun_out/ring/pytest_pp_all.log 2>&1
echo "rc $?" >> gpurun_out/ring/pytest_pp_all.log
tail -6 gpurun_out/ring/pytest_pp_all.log
