#!/bin/bash
# GPU box: KR / Louvain / full-size / big tests and the bench lines after the stream-policy change
out=gpurun_out/${1:-ntc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_kr.py tests/test_gpu_louvain.py tests/test_gpu_fullsize.py tests/test_gpu_big.py tests/test_gpu_pipeline.py -m gpu -x -q --durations=5 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -9 $out/tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py | head -5
done
