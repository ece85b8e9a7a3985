out=gpurun_out/r4_t6
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $out/pytest_all.log 2>&1
tail -8 $out/pytest_all.log
for v in "B3C_X=1" "B3C_LV_NO_AGG_HASH=1"; do
  echo "== variant [$v] C3 survey"
  env $v timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
done
echo "== C3 default"
timeout 400 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-survey-leg 2>/dev/null | tail -1 | python tools/bsum.py
for v in "B3C_KR_TCD_R=256" "B3C_KR_TCD_R=128" "B3C_KR_TCD_DCAP=2560"; do
  echo "== variant [$v] C3 survey"
  env $v timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
done
timeout 900 python tools/fuzz_ill.py 3000 7 > $out/fuzz_ill.log 2>&1
tail -6 $out/fuzz_ill.log
