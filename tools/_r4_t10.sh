out=gpurun_out/r4_t10
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python -u -m pytest tests -q -m gpu --durations=20 -p no:cacheprovider ) > $out/pytest_all.log 2>&1
tail -32 $out/pytest_all.log
