#!/bin/bash
# GPU box: the whole `-m gpu` suite as the driver runs it, with the slowest tests listed.
# usage (from the build container): tools/gpu_retry.sh 1800 'bash tools/gpu_suite.sh <tag>'
out=gpurun_out/${1:-suite}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu ${B3C_SUITE_FLAGS--x} -q --durations=12 > $out/tests_gpu.log 2>&1
echo "tests rc $?" >> $out/tests_gpu.log
tail -18 $out/tests_gpu.log
