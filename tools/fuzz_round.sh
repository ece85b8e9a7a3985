#!/bin/bash
# GPU box: __graft_entry__.smoke() and the round's differential campaign on the final tree -- tools/fuzz_parity.py under the defaults,
# with the tail pass forced on every eligible map, with the sort aggregation forced on every Louvain level, with every dictionary
# dropped, and with hub rows cut into pieces of 256 entries + the non-temporal stream policy forced on every map.
# usage (from the build container): tools/gpu_retry.sh 3000 'bash tools/fuzz_round.sh <tag> <cases per seed> [<seed base>]'
out=gpurun_out/${1:-fuzz}
cases=${2:-500}
base=${3:-500}            # seeds base+1 .. base+5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log
timeout 1500 python tools/fuzz_parity.py $cases $((base+1)) > $out/fuzz_default.log 2>&1; echo "rc $?" >> $out/fuzz_default.log
B3C_KR_TCD_XT=1 timeout 1500 python tools/fuzz_parity.py $cases $((base+2)) > $out/fuzz_tails.log 2>&1; echo "rc $?" >> $out/fuzz_tails.log
B3C_LV_AGG_SORT=1 B3C_LV_NO_AGG_HASH=1 timeout 1500 python tools/fuzz_parity.py $cases $((base+3)) > $out/fuzz_aggsort.log 2>&1; echo "rc $?" >> $out/fuzz_aggsort.log
B3C_KR_TCD_XT=1 B3C_KR_TCD_MINUSE8=100 timeout 1500 python tools/fuzz_parity.py $cases $((base+4)) > $out/fuzz_nodict.log 2>&1; echo "rc $?" >> $out/fuzz_nodict.log
B3C_KR_HUB_SPLIT=256 B3C_KR_NT_MIN=0 B3C_LV_NT_MIN=0 timeout 1500 python tools/fuzz_parity.py $cases $((base+5)) > $out/fuzz_hubpieces.log 2>&1; echo "rc $?" >> $out/fuzz_hubpieces.log
for f in smoke fuzz_default fuzz_tails fuzz_aggsort fuzz_nodict fuzz_hubpieces; do echo "== $f"; tail -3 $out/$f.log; done
