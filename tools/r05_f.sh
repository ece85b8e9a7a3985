#!/bin/bash
out=gpurun_out/r05f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_full_size_matches_the_oracle" tests/test_gpu_kr.py -m gpu -x -q --durations=5 > $out/tests.log 2>&1
echo "tests rc $?" >> $out/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
find $out/trace_c3 -name "*kernel_stats.csv" -exec cp {} $out/trace_c3_kernel_stats.csv \;
rm -rf $out/trace_c3
tail -4 $out/tests.log
for f in bench_c3 bench_c2; do tail -1 $out/$f.json | python tools/bsum.py; done
python tools/kstats.py $out/trace_c3_kernel_stats.csv 16
