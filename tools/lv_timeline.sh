#!/bin/bash
# GPU box: kernel trace of two C3 steps (contract workload), reduced to a per-sweep table of the Louvain stage.
# usage: tools/gpu_retry.sh 900 'bash tools/lv_timeline.sh <tag> [config]'
out=gpurun_out/${1:-lvt}
cfg=${2:-C3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B3C_LV_DEBUG=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/bench.json 2> $out/bench.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/lv_timeline.py $f > $out/timeline.txt 2>&1
grep '\[b3c\]' $out/bench.err | tail -80 > $out/lv_debug.txt
rm -rf $out/trace
tail -1 $out/bench.json | python tools/bsum.py
cat $out/timeline.txt | head -150
