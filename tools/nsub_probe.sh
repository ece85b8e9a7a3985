#!/bin/bash
# GPU box: what the number of sub-rounds per Louvain sweep does to the contract workload (C3, C2): time, sweeps, levels, communities, modularity
out=gpurun_out/${1:-nsub}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in C3 C2; do
for ns in 4 8 16 32 64; do
  B3C_KR_COOP=0 B3C_LV_DEBUG=1 timeout 300 python bench.py --config $cfg --n-sub $ns --steps 3 --warmup 1 --no-cpu-baseline --no-variant-leg > $out/bench_${cfg}_ns$ns.json 2> $out/bench_${cfg}_ns$ns.err
  tail -1 $out/bench_${cfg}_ns$ns.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$cfg n_sub $ns: %.2f ms/step louvain %.2f ms sweeps %d levels %d comm0 %d comm %d Q %.6f' % (d['ms_per_step'], c['stage_ms']['louvain'], c['louvain_sweeps_per_step'], c['louvain_levels'], c['communities_level0'], c['communities'], c['modularity']))"
  grep '\[b3c\] level' $out/bench_${cfg}_ns$ns.err | tail -40 | awk '/nodes,/{print}' | tail -5
done
done
