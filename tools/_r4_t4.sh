out=gpurun_out/r4_t4
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kr.py tests/test_gpu_fuzz.py tests/test_dist.py -x -q -m gpu > $out/pytest.log 2>&1
tail -15 $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_big.py -x -q -m gpu -k stated > $out/pytest_c5.log 2>&1
tail -3 $out/pytest_c5.log
for v in "B3C_X=1" "B3C_KR_XT_PACK=0" "B3C_LV_WAVE_MAX=256" "B3C_KR_XT_S=12288"; do
  echo "== variant [$v] C3 survey"
  env $v B3C_KR_DEBUG=1 timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>$out/err_$v.txt | tail -1 | python tools/bsum.py
  grep "tail pass" $out/err_$v.txt | head -3
  echo "== variant [$v] C3"
  env $v timeout 400 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-survey-leg 2>/dev/null | tail -1 | python tools/bsum.py
done
