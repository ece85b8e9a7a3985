out=gpurun_out/r04f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python -u -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider ) > $out/pytest_all.log 2>&1
tail -16 $out/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-survey-leg 2>/dev/null | tail -1 | python tools/bsum.py
