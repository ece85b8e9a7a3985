#!/bin/bash
# GPU box: quick Louvain-side check after a kernel change: Louvain tests, the C2 / C3 oracle comparisons on the contract law, one C3 bench line
out=gpurun_out/${1:-lvq}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_power_law_map_matches_the_oracle" "tests/test_gpu_fullsize.py::test_full_size_matches_the_oracle[C2]" "tests/test_gpu_fullsize.py::test_full_size_matches_the_oracle[C3]" -m gpu -x -q --durations=6 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -10 $out/tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py | head -3
done
timeout 600 python bench.py --config C2 --steps 10 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c2.json 2> $out/bench_c2.err
tail -1 $out/bench_c2.json | python tools/bsum.py | head -3
