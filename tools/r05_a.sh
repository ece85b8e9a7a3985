#!/bin/bash
# round 5, first GPU call: the new tests, the bench line in the driver's form, diagnostics of the contract workload
out=gpurun_out/r05a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REGEX="kr_kb_vec|kr_kb_tails|kr_kb_tcd2|kr_prep_counts|k_lv_decide_g|k_lv_first_g"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_dist.py tests/test_gpu_louvain.py tests/test_gpu_edges.py -m gpu -x -q --durations=15 -s > $out/tests.log 2>&1
echo "tests rc $?" >> $out/tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $out/bench_c3.json 2> $out/bench_c3.err
B3C_LV_DEBUG=1 B3C_KR_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/diag.json 2> $out/diag.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 3 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_write -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_write.log 2>&1
find $out/trace_c3 -name "*kernel_stats.csv" -exec cp {} $out/trace_c3_kernel_stats.csv \;
rm -rf $out/trace_c3
tail -3 $out/tests.log
tail -1 $out/bench_c3.json | python tools/bsum.py
tail -5 $out/bench_c3.err
