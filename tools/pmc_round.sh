#!/bin/bash
# GPU box: the counter passes of tools/profile_round.sh by themselves (HBM traffic of the SpMV kernels and the Louvain scan on the
# contract map, SQ / TA / TCP groups of the tail pass and of the scan, the SpMV forms side by side) -- after a late kernel change.
# usage: tools/gpu_retry.sh 2400 'bash tools/pmc_round.sh <tag>'; then python tools/traffic_reduce.py gpurun_out/<tag> r05 C3
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REGEX="kr_kb_vec|kr_kb_tails|kr_kb_tcd2|kr_kb_epi|kr_prep_counts|k_lv_decide_g|k_lv_first_g"
timeout 600 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_fetch -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$REGEX" --output-format csv -d $out/pmc_write -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_write.log 2>&1
B3C_KR_TCD_TLOG=1 timeout 300 python tools/tcd_probe.py C3 "" "B3C_KR_TCD_ALLTAILS=0" "B3C_KR_NO_TCD=1" > $out/tcd_timeline.txt 2>&1
B3C_PMC_KERNEL="kr_kb_tails" bash tools/pmc_kr.sh $tag/tails > $out/pmc_tails.log 2>&1
python tools/pmc_reduce.py $out/tails > $out/kr_kb_tails_counters.txt 2>&1
grep -E "^\(default\)|^B3C_|vs first" $out/tcd_timeline.txt | cut -c1-200
tail -1 $out/bench_c3.json | python tools/bsum.py
