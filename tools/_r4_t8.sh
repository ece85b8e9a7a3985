out=gpurun_out/r4_t8
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python -u -m pytest tests/test_gpu_louvain.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -q -m gpu --durations=25 --timeout=500 -p no:cacheprovider ) > $out/pytest_a.log 2>&1
tail -45 $out/pytest_a.log
