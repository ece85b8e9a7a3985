#!/bin/bash
# GPU box: layout parameters of the tail pass / dictionary rule on the contract map (tools/tcd_probe.py variants)
# usage: tools/gpu_retry.sh 1200 'bash tools/xt_probe.sh <tag> "<variant>" "<variant>" ...'   (default: super-tile size / chunk count)
out=gpurun_out/${1:-xt}
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $# -eq 0 ]; then set -- "" "B3C_KR_XT_S=12288" "B3C_KR_XT_S=12288,B3C_KR_XT_CHUNKS=3" "B3C_KR_XT_S=12288,B3C_KR_XT_CHUNKS=4" "B3C_KR_XT_CHUNKS=3" "B3C_KR_XT_CHUNKS=4"; fi
timeout 900 python tools/tcd_probe.py C3 "$@" 2>&1 | grep -v "SpMV launch of tick" > $out/xt_probe.txt
grep -E "^\(default\)|^B3C_|vs first" $out/xt_probe.txt | cut -c1-200
