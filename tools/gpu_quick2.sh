#!/bin/bash
# GPU box: chosen test files + one C3 bench line + kernel stats (development helper).  usage: gpu_quick2.sh <tag> <pytest args...>
out=gpurun_out/${1:-q2}
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest "$@" -m gpu -x -q --durations=5 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -8 $out/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c3.json 2> $out/bench_c3.err
tail -1 $out/bench_c3.json | python tools/bsum.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
find $out/trace_c3 -name "*kernel_stats.csv" -exec cp {} $out/trace_c3_kernel_stats.csv \;
rm -rf $out/trace_c3
python tools/kstats.py $out/trace_c3_kernel_stats.csv 30 | grep -i "compress\|permute\|total kernel"
