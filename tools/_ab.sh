#!/bin/bash
mkdir -p gpurun_out/ab
timeout 500 python -m pytest tests/test_gpu_louvain.py -x -q -m gpu --durations=3 > gpurun_out/ab/lv.log 2>&1
for v in 1 0 1 0; do
  B3C_LV_WIDE_G=$v timeout 200 python bench.py --config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-survey-leg 2>/dev/null | tail -1 | python tools/bsum.py > gpurun_out/ab/c3_w$v.txt.$RANDOM
done
B3C_LV_WIDE_G=1 timeout 200 python bench.py --config C3 --synth survey --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py > gpurun_out/ab/sv_w1.txt
B3C_LV_WIDE_G=0 timeout 200 python bench.py --config C3 --synth survey --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/bsum.py > gpurun_out/ab/sv_w0.txt
tail -3 gpurun_out/ab/lv.log; head -50 gpurun_out/ab/*.txt*
