#!/bin/bash
# GPU box: a handful of tests + one bench line (development helper)
out=gpurun_out/${1:-quick}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kr.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -15 $out/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py
