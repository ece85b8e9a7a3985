out=gpurun_out/r4_t7
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -q -m gpu ) > $out/pytest_all.log 2>&1
tail -12 $out/pytest_all.log
