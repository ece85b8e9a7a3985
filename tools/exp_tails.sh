#!/bin/bash
# GPU box: quick A/B (development): a few suites + bench lines
out=gpurun_out/exp_tails
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_golden2.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for i in 1 2; do
timeout 300 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py
done
timeout 300 python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
tail -1 $out/bench_c2.json | python tools/bsum.py
