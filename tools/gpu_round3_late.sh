#!/bin/bash
# late round-3 checks: the bench watchdog, eight ranks on one GPU (gloo transport), the default bench line
set -u
mkdir -p gpurun_out/late
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_pipeline.py -x -q -m gpu -k "deadline or strong_scaling" > gpurun_out/late/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/late/pytest.log
tail -5 gpurun_out/late/pytest.log
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 8 \
    --steps 5 --warmup 2 --backend gloo --no-cpu-baseline --s3-bins 400000 > gpurun_out/late/bench8_gloo.log 2>&1
echo "bench8 rc $?" >> gpurun_out/late/bench8_gloo.log
grep -c '^{' gpurun_out/late/bench8_gloo.log
tail -c 3000 gpurun_out/late/bench8_gloo.log
timeout 900 python bench.py > gpurun_out/late/bench1.json 2> gpurun_out/late/bench1.err
echo "bench1 rc $?"
tail -c 1500 gpurun_out/late/bench1.json
