out=gpurun_out/r4_t2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_louvain.py tests/test_gpu_kr.py -x -q -m gpu > $out/pytest.log 2>&1
tail -3 $out/pytest.log
for v in "" "B3C_LV_NO_XCD=1"; do
  echo "== variant [$v] C3 survey"
  env $v timeout 400 python bench.py --config C3 --synth survey --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py
  echo "== variant [$v] C3"
  env $v timeout 400 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-survey-leg 2>/dev/null | tail -1 | python tools/bsum.py
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_lv_decide_g|k_lv_first_g" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-survey-leg > $out/pmc_fetch.log 2>&1
find $out/pmc_fetch -name "*counter_collection.csv" -exec cp {} $out/c3_lv_FETCH.csv \;
rm -rf $out/pmc_fetch
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4_t2/c3_lv_FETCH.csv')))
for r in rows[:16]:
    print(r['Kernel_Name'][:40], r['Counter_Name'], r['Counter_Value'])
PY
