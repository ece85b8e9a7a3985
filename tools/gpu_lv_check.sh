#!/bin/bash
# GPU box: Louvain-side check after a kernel change: Louvain / edges / fuzz tests, the C5 contract test, one C5 bench line
out=gpurun_out/${1:-lvc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_fuzz.py tests/test_gpu_big.py "tests/test_gpu_fullsize.py::test_power_law_map_matches_the_oracle" "tests/test_gpu_fullsize.py::test_full_size_matches_the_oracle[C2]" -m gpu -x -q --durations=6 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -10 $out/tests.log
timeout 900 python bench.py --config C5 --steps 1 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
tail -1 $out/bench_c5.json | python tools/bsum.py
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py
