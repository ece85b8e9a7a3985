#!/bin/bash
# usage: tools/gpu_retry.sh 1500 'bash tools/c5_probe.sh <tag> [max_iter] "<variant>" ...'
out=gpurun_out/${1:-c5p}
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
it=${1:-60}
shift
if [ $# -eq 0 ]; then set -- "" "B3C_KR_XT_CHUNKS=8" "B3C_KR_XT_CHUNKS=24" "B3C_KR_XT_CHUNKS=32" "B3C_KR_XT_CHUNKS=48" "B3C_KR_XT_S=8192" "B3C_KR_XT_S=8192,B3C_KR_XT_CHUNKS=32" "B3C_KR_TCD_ALLTAILS=0" "B3C_KR_NO_TCD=1"; fi
timeout 1400 python tools/c5_probe.py $it "$@" > $out/c5_probe.txt 2>&1
grep -E "^C5|^\(default\)|^B3C_|vs first|Error|error" $out/c5_probe.txt | cut -c1-260
