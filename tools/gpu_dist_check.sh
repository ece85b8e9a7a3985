#!/bin/bash
# GPU box: the sharded tests (two / three ranks on one GPU) by themselves
out=gpurun_out/${1:-dist}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dist.py -m gpu -q --durations=8 > $out/tests_dist.log 2>&1; echo "rc $?" >> $out/tests_dist.log
tail -14 $out/tests_dist.log
