#!/bin/bash
# GPU box: quick A/B (development): a few suites + bench lines
out=gpurun_out/exp_tails
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1400 python tools/c5_probe.py 5e6 1e9 '' B3C_KR_XT_S=8192 B3C_KR_XT_CHUNKS=8 B3C_KR_XT_CHUNKS=24 2>&1 | grep -v 'tick' > $out/c5.txt; grep -v '^\[b3c\]' $out/c5.txt
timeout 600 python tools/c5_probe.py 2e6 3e8 '' B3C_KR_XT_S=8192 B3C_KR_XT_CHUNKS=2 B3C_KR_XT_S=8192,B3C_KR_XT_CHUNKS=2 2>&1 | grep -v 'tick' > $out/c5b.txt; grep -v '^\[b3c\]' $out/c5b.txt
