#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ring
timeout 900 python tools/s3_ring_ab.py --reps 2 --variants "3::,4:pp:,4:pn:,4:pl:,3::,4:pp:,4:pn:,4:pl:" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ring/ab_prio.log
for v in pn pl; do
EPG_S3_SYRK=$v EPG_S3_RING=4 EPG_S3_DBG=32 timeout 600 python tools/s3_ring_ab.py --child --reps 0 --bins 1048576 > gpurun_out/ring/trace_$v.log 2>&1
echo "---- $v"
grep "trace w0" gpurun_out/ring/trace_$v.log | sed -n 3,9p
grep "trace w4" gpurun_out/ring/trace_$v.log | sed -n 3,9p
done
