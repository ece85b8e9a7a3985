#!/bin/bash
# GPU box: super-tile size / chunk count of the tail pass on the contract map (tools/tcd_probe.py variants)
out=gpurun_out/${1:-xt}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/tcd_probe.py C3 "" "B3C_KR_XT_S=12288" "B3C_KR_XT_S=12288,B3C_KR_XT_CHUNKS=3" "B3C_KR_XT_S=12288,B3C_KR_XT_CHUNKS=4" "B3C_KR_XT_CHUNKS=3" "B3C_KR_XT_CHUNKS=4" 2>&1 | grep -v "SpMV launch of tick" > $out/xt_probe.txt
cat $out/xt_probe.txt | cut -c1-200
