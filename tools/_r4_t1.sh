out=gpurun_out/r4_t1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_louvain.py -x -q -m gpu > $out/pytest_louvain.log 2>&1
tail -5 $out/pytest_louvain.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "other_workload or C2" > $out/pytest_full.log 2>&1
tail -5 $out/pytest_full.log
B3C_LV_DEBUG=1 timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_survey.json 2> $out/bench_c3_survey.err
tail -1 $out/bench_c3_survey.json | python tools/bsum.py
B3C_LV_NO_MEGA_LDS=1 timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
timeout 400 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
timeout 400 python bench.py --config C2 --synth survey --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3s -- python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline > $out/trace_c3s.log 2>&1
find $out/trace_c3s -name "*kernel_stats.csv" -exec cp {} $out/c3s_kernel_stats.csv \;
rm -rf $out/trace_c3s
head -30 $out/c3s_kernel_stats.csv | cut -c1-160
