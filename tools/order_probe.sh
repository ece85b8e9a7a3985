#!/bin/bash
out=gpurun_out/${1:-order}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/order_probe.py C3 survey > $out/order_c3.txt 2>&1
timeout 600 python tools/order_probe.py C3 sqrt > $out/order_c3_variant.txt 2>&1
tail -3 $out/order_c3.txt; tail -3 $out/order_c3_variant.txt
