out=gpurun_out/r4_t9
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 600 python -u -m pytest tests/test_gpu_louvain.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -x -q -m gpu --durations=12 -p no:cacheprovider ) > $out/pytest_a.log 2>&1
tail -30 $out/pytest_a.log
for v in "B3C_X=1" "B3C_LV_NO_MEGA_SPLIT=1"; do
  echo "== variant [$v] C3 survey"
  env $v timeout 300 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
done
echo "== C2, C2 survey"
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
