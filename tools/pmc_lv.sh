#!/bin/bash
# GPU box: counter groups over the Louvain grouped scan, reduced per LAUNCH (first sweep against late sweeps)
tag=${1:-pmclv}
B3C_PMC_KERNEL="k_lv_decide_g" bash tools/pmc_kr.sh $tag > gpurun_out/$tag.log 2>&1
python tools/pmc_by_launch.py gpurun_out/$tag 'k_lv_decide_g<64, 4>' 64 > gpurun_out/$tag/by_launch.txt
cat gpurun_out/$tag/by_launch.txt
find gpurun_out/$tag -name "*.csv" -size +3M -delete
