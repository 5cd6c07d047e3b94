#!/bin/bash
set -u
mkdir -p gpurun_out/ring
cd "$GRAFT_REPO_ROOT"
timeout 1200 python tools/s3_ring_ab.py --reps 2 --variants "${1:-3::,4:pp:}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ring/ab.log
