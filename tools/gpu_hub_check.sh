#!/bin/bash
# GPU box: KR tests + the C5 probe (default form only) after a change of the hub kernel
out=gpurun_out/${1:-hub}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kr.py tests/test_gpu_edges.py -m gpu -x -q --durations=5 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -8 $out/tests.log
timeout 900 python tools/c5_probe.py 60 "" "B3C_KR_HUB_SPLIT=4096" "B3C_KR_HUB_SPLIT=65536" "B3C_KR_HUB_SPLIT=100000000" > $out/c5_probe.txt 2>&1
grep -E "^C5|^\(default\)|^B3C_|vs first|Error|error" $out/c5_probe.txt | cut -c1-260
