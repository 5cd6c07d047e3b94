#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w4
timeout 1500 python tools/s3_ring_ab.py --reps 2 --variants "$1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/w4/ab.log
