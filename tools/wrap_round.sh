#!/bin/bash
# GPU box, last call of a round: the whole -m gpu suite, smoke() and the differential campaign on the final tree.
# usage (from the build container): tools/gpu_retry.sh 4000 'bash tools/wrap_round.sh <tag> <fuzz cases per seed>'
bash tools/gpu_suite.sh $1
bash tools/fuzz_round.sh $1 ${2:-500}
