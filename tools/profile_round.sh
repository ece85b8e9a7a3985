#!/bin/bash
# GPU box: the measurements behind profiles/ for one round -- bench lines, rocprofv3 kernel-trace stats (C3 and C2), HBM
# traffic counters of the dominant kernel (separate --pmc passes, with two pure streaming kernels of known byte counts in
# the same passes as calibration), and the per-tile timeline of the dictionary SpMV.
# Every command runs under its own timeout: a profiler that hangs after a fault would otherwise eat the GPU budget.
# usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --config C3 --steps 5 --warmup 2 > $out/bench_c3.json 2> $out/bench_c3.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $out/trace_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -- python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > $out/trace_c2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "kr_kb_vec|kr_kb_tails|kr_kb_tcd2|kr_prep_counts|k_lv_decide_g" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "kr_kb_vec|kr_kb_tails|kr_kb_tcd2|kr_prep_counts|k_lv_decide_g" --output-format csv -d $out/pmc_write -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.log 2>&1
B3C_KR_TCD_TLOG=1 timeout 300 python tools/tcd_probe.py C3 "" "B3C_KR_TCD_XT=0,B3C_KR_TCD_DCAP=4096" "B3C_KR_NO_TCD=1" > $out/tcd_timeline.txt 2>&1
# the surveyed workload definition and a hard clustering problem beside the headline
timeout 400 python bench.py --config C3 --synth survey --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3_survey.json 2> $out/bench_c3_survey.err
timeout 300 python bench.py --config C2 --hard --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2_hard.json 2> $out/bench_c2_hard.err
timeout 300 python bench.py --config C2 --synth survey --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2_survey.json 2> $out/bench_c2_survey.err
# VGPRs / scratch / LDS / occupancy of every kernel as compiled (a stack object in scratch shows up here, not in a timing)
timeout 900 python tools/resource_usage.py > $out/resource_usage.txt 2>&1
tail -1 $out/bench_c3.json | python tools/bsum.py
tail -1 $out/bench_c2.json | python tools/bsum.py
tail -1 $out/bench_c3_survey.json | python tools/bsum.py
tail -1 $out/bench_c2_hard.json | python tools/bsum.py
tail -1 $out/bench_c2_survey.json | python tools/bsum.py
