#!/bin/bash
# GPU box: BASELINE.json configs[4] on ONE GPU -- 5M contigs, 1e9 UNIQUE power-law edges (drawn on the device until the count is
# reached), the whole cluster step; then the HBM traffic counters of the SpMV pair on the same map (separate --pmc passes).
# usage: tools/run_c5.sh <tag>     (outputs under gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py --config C5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
tail -1 $out/bench_c5.json | python tools/bsum.py
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c5 -- python bench.py --config C5 --steps 1 --warmup 0 > $out/trace_c5.log 2>&1
timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "kr_kb_tails|kr_kb_tcd2|kr_kb_hub" --output-format csv -d $out/pmc_fetch -- python bench.py --config C5 --steps 1 --warmup 0 > $out/pmc_fetch.log 2>&1
timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "kr_kb_tails|kr_kb_tcd2|kr_kb_hub" --output-format csv -d $out/pmc_write -- python bench.py --config C5 --steps 1 --warmup 0 > $out/pmc_write.log 2>&1
ls -la $out/*
