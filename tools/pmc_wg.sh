#!/bin/bash
# GPU box: counter groups over the workgroup-per-node scan of Louvain's later levels (k_lv_decide_wg), reduced per LAUNCH
tag=${1:-pmcwg}
B3C_PMC_KERNEL="k_lv_decide_wg" bash tools/pmc_kr.sh $tag > gpurun_out/$tag.log 2>&1
python tools/pmc_by_launch.py gpurun_out/$tag 'k_lv_decide_wg<11, 256>' 32 > gpurun_out/$tag/by_launch.txt
cat gpurun_out/$tag/by_launch.txt | cut -c1-260
find gpurun_out/$tag -name "*.csv" -size +3M -delete
