#!/bin/bash
# GPU box: HBM-side traffic of the KR SpMV's kernels at C5 on the contract generator (separate --pmc passes, one step each)
# usage: tools/gpu_retry.sh 1500 'bash tools/pmc_c5.sh <tag>'; then python tools/traffic_reduce.py gpurun_out/<tag> r06 C5
out=gpurun_out/${1:-pmcc5}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REGEX="kr_kb_tails|kr_kb_tcd2|kr_kb_epi|kr_kb_hub|kr_prep_counts"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_fetch -- python bench.py --config C5 --steps 1 --warmup 0 > $out/bench_c5.json 2> $out/pmc_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_write -- python bench.py --config C5 --steps 1 --warmup 0 > $out/bench_c5_w.json 2> $out/pmc_write.log
ls -la $out/pmc_fetch/*/ | tail -3
tail -1 $out/bench_c5.json | python tools/bsum.py
