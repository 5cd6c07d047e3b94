L=${1:-r03b}
mkdir -p gpurun_out/$L
python -m pytest tests -m gpu -x -q > gpurun_out/$L/tests.log 2>&1; tail -15 gpurun_out/$L/tests.log
for extra in "" "--pg" "--pg --graph"; do
  python bench.py --bins 1875000 --steps 200 --warmup 20 --no-cpu-baseline --configs none --placement-experiment 0 $extra > gpurun_out/$L/shard$(echo $extra | tr -d ' -').json 2>> gpurun_out/$L/shard.err
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --configs none --placement-experiment 0 > gpurun_out/$L/full.json 2>> gpurun_out/$L/shard.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --configs none --placement-experiment 0 --pg --graph > gpurun_out/$L/fullpggraph.json 2>> gpurun_out/$L/shard.err
for f in gpurun_out/$L/*.json; do echo $f; python tools/bench_digest.py < $f; done
