#!/bin/bash
# GPU box: bench line + rocprofv3 kernel-trace stats + HBM traffic counters for the dominant kernel.
# usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --config C3 --steps 3 --warmup 1 > $out/bench_c3.json 2> $out/bench_c3.err
python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "kr_kb_vec" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "kr_kb_vec" --output-format csv -d $out/pmc_write -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.log 2>&1
tail -1 $out/bench_c3.json | python tools/bsum.py
tail -1 $out/bench_c2.json | python tools/bsum.py
