#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in torch own; do
  B3C_PROBE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$mode -- python tools/kr_counts_probe.py C2 > gpurun_out/ab_$mode.log 2>&1
  echo "== $mode"; tail -1 gpurun_out/ab_$mode.log; python tools/kstats.py gpurun_out/ab_$mode 5
done
