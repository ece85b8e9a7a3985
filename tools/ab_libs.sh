#!/bin/bash
# GPU box (development helper): A/B of prebuilt library variants under ab/*.so on one box -- each copied over bin3c_amd/libb3c.so in turn
# usage: tools/gpu_retry.sh 1200 'bash tools/ab_libs.sh <tag> [bench args]'
out=gpurun_out/${1:-ab}
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp bin3c_amd/libb3c.so $out/.orig.so
for rep in 1 2; do
for f in ab/*.so; do
  cp $f bin3c_amd/libb3c.so
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variant-leg "$@" > $out/b.json 2> $out/b.err
  echo "== $f"; tail -1 $out/b.json | python tools/bsum.py | head -4
done
done
cp $out/.orig.so bin3c_amd/libb3c.so; rm -f $out/.orig.so
