#!/bin/bash
# GPU box: KR tests, bench C3 + C2 lines with the KR epilogue inside the tail pass and as its own launch, kernel stats of C3
out=gpurun_out/${1:-krc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kr.py -m gpu -x -q --durations=5 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -12 $out/tests.log
for sw in 1 0; do
  B3C_KR_EPI_IN_TAILS=$sw timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/bench_c3_epi$sw.json 2> $out/bench_c3_epi$sw.err
  tail -1 $out/bench_c3_epi$sw.json | python tools/bsum.py
  B3C_KR_EPI_IN_TAILS=$sw timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-variant-leg > $out/bench_c2_epi$sw.json 2> $out/bench_c2_epi$sw.err
  tail -1 $out/bench_c2_epi$sw.json | python tools/bsum.py
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
find $out/trace_c3 -name "*kernel_stats.csv" -exec cp {} $out/trace_c3_kernel_stats.csv \;
rm -rf $out/trace_c3
python tools/kstats.py $out/trace_c3_kernel_stats.csv 12
