#!/bin/bash
# GPU box: the non-temporal policy of the matrix / entry streams (KR_STREAM_NT_MIN, LV_STREAM_NT_MIN) against the default policy
# (B3C_KR_NT_MIN=0x4000000000000000 B3C_LV_NT_MIN=0x4000000000000000) on the contract C3 / C2, the variant C3 and the C5 contract SpMV
out=gpurun_out/${1:-ntp}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kr.py tests/test_gpu_louvain.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -3 $out/tests.log
for args in "" "--synth sqrt" "--config C2" "--config C2 --synth sqrt"; do
for rep in 1 2; do
for sw in "" "B3C_KR_NT_MIN=0x4000000000000000 B3C_LV_NT_MIN=0x4000000000000000"; do
  env $sw timeout 600 python bench.py $args --steps 10 --warmup 2 --no-cpu-baseline --no-variant-leg > $out/b.json 2> $out/b.err
  echo "== [$args] [$sw]"; tail -1 $out/b.json | python tools/bsum.py | head -3 | tail -2
done
done
done
timeout 600 python tools/c5_probe.py 60 "" "B3C_KR_NT_MIN=0x4000000000000000" 2>&1 | grep -E "^\(default\)|^B3C_|vs first" | cut -c1-200
