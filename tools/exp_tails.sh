#!/bin/bash
# GPU box: quick A/B (development): a few suites + bench lines
out=gpurun_out/exp_tails
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 300 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $out/trace_c3.log 2>&1
