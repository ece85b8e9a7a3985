#!/bin/bash
# GPU box, after tools/profile_round.sh + tools/traffic_reduce.py: the bench lines once more with this build's counter records on file
# (roofline.traffic is read from profiles/traffic.json as committed before the run), the C5 lines and a C5 kernel trace.
# usage (from the build container): tools/gpu_retry.sh 3000 'bash tools/final_round.sh <tag>'
out=gpurun_out/${1:-final}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --config C3 --steps 5 --warmup 2 > $out/bench_c3.json 2> $out/bench_c3.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3_driver_form.json 2> $out/bench_c3_driver_form.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
timeout 900 python bench.py --config C5 --c5-counts unique --steps 2 --warmup 1 > $out/bench_c5_unique.json 2> $out/bench_c5_unique.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c5 -- python bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline > $out/trace_c5.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c5_unique -- python bench.py --config C5 --c5-counts unique --steps 1 --warmup 0 --no-cpu-baseline > $out/trace_c5_unique.log 2>&1
for t in trace_c3 trace_c5 trace_c5_unique; do
  find $out/$t -name "*kernel_stats.csv" -exec cp {} $out/${t}_kernel_stats.csv \;
  rm -rf $out/$t
done
for f in bench_c3 bench_c3_driver_form bench_c5_unique; do tail -1 $out/$f.json | python tools/bsum.py; done
python tools/kstats.py $out/trace_c3_kernel_stats.csv 12
python tools/kstats.py $out/trace_c5_kernel_stats.csv 10
python tools/kstats.py $out/trace_c5_unique_kernel_stats.csv 6
