#!/bin/bash
# GPU box: the measurements behind profiles/ for one round -- bench lines (headline = the workload as SURVEY 8(d) defines it, + the rank^-0.5 variant + C2 + hard + C5), rocprofv3
# kernel-trace stats (C3, C3 on the variant law, C2, C5), HBM traffic counters of the SpMV pair and of the Louvain scan (separate
# --pmc passes, with a pure streaming kernel of known byte count in the same passes as calibration), SQ / TA / TCP counter groups
# of the Louvain scan, the per-tile timeline of the dictionary SpMV and the resource usage of every kernel as compiled.
# Every command runs under its own timeout: a profiler that hangs after a fault would otherwise eat the GPU budget.
# usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/; tools/traffic_reduce.py and a copy into profiles/ follow)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REGEX="kr_kb_vec|kr_kb_tails|kr_kb_tcd2|kr_kb_epi|kr_prep_counts|k_lv_decide_g|k_lv_first_g"
timeout 900 python bench.py --config C3 --steps 5 --warmup 2 > $out/bench_c3.json 2> $out/bench_c3.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
# the line in the form the driver runs it
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3_driver_form.json 2> $out/bench_c3_driver_form.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/trace_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3_variant -- python bench.py --config C3 --synth sqrt --steps 3 --warmup 1 --no-cpu-baseline > $out/trace_c3_variant.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -- python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-variant-leg > $out/trace_c2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_write -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_write.log 2>&1
B3C_KR_TCD_TLOG=1 timeout 300 python tools/tcd_probe.py C3 "" "B3C_KR_TCD_ALLTAILS=0" "B3C_KR_NO_TCD=1" > $out/tcd_timeline.txt 2>&1
# the rank^-0.5 genome-size variant (the headline of rounds 1-4) and a hard clustering problem as their own lines
timeout 400 python bench.py --config C3 --synth sqrt --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3_variant.json 2> $out/bench_c3_variant.err
timeout 300 python bench.py --config C2 --hard --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2_hard.json 2> $out/bench_c2_hard.err
timeout 300 python bench.py --config C2 --synth sqrt --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c2_variant.json 2> $out/bench_c2_variant.err
# BASELINE's largest map on ONE GPU: the contract generator (duplicates summed) and the stated number of unique edges
timeout 900 python bench.py --config C5 --steps 1 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
timeout 900 python bench.py --config C5 --c5-counts unique --steps 2 --warmup 1 > $out/bench_c5_unique.json 2> $out/bench_c5_unique.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c5 -- python bench.py --config C5 --c5-counts unique --steps 1 --warmup 0 > $out/trace_c5.log 2>&1
# SQ / TA / TCP counter groups of the Louvain neighbour scan
B3C_PMC_KERNEL="k_lv_decide_g" bash tools/pmc_kr.sh $tag/lv > $out/pmc_lv.log 2>&1
python tools/pmc_reduce.py $out/lv > $out/k_lv_decide_g_counters.txt 2>&1
# ... the same per LAUNCH, in launch order (first sweep against the pruned late sweeps)
python tools/pmc_by_launch.py $out/lv 'k_lv_decide_g<64, 4>' 64 > $out/k_lv_decide_g_by_launch.txt 2>&1
# ... and of the tail pass of the SpMV
B3C_PMC_KERNEL="kr_kb_tails" bash tools/pmc_kr.sh $tag/tails > $out/pmc_tails.log 2>&1
python tools/pmc_reduce.py $out/tails > $out/kr_kb_tails_counters.txt 2>&1
# VGPRs / scratch / LDS / occupancy of every kernel as compiled (a stack object in scratch shows up here, not in a timing)
timeout 900 python tools/resource_usage.py > $out/resource_usage.txt 2>&1
for f in bench_c3 bench_c2 bench_c3_variant bench_c2_hard bench_c2_variant bench_c5 bench_c5_unique; do tail -1 $out/$f.json | python tools/bsum.py; done
# keep the merge small: kernel stats only
for t in trace_c3 trace_c3_variant trace_c2 trace_c5; do
  find $out/$t -name "*kernel_stats.csv" -exec cp {} $out/${t}_kernel_stats.csv \;
  rm -rf $out/$t
done
