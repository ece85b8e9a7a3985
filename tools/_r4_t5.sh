out=gpurun_out/r4_t5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_big.py -x -q -m gpu > $out/pytest.log 2>&1
tail -8 $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "other_workload or power_law" > $out/pytest_full.log 2>&1
tail -5 $out/pytest_full.log
for v in "B3C_X=1" "B3C_LV_NO_AGG_HASH=1"; do
  echo "== variant [$v] C3 survey"
  env $v timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
done
echo "== C5 unique"
B3C_LV_DEBUG=1 timeout 900 python bench.py --config C5 --c5-counts unique --steps 1 --warmup 1 2>$out/c5.err | tail -1 | python tools/bsum.py
grep "level" $out/c5.err | head -20
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3s -- python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline > $out/trace_c3s.log 2>&1
find $out/trace_c3s -name "*kernel_stats.csv" -exec cp {} $out/c3s_kernel_stats.csv \;
rm -rf $out/trace_c3s
