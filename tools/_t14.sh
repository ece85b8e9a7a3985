cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 900 python -u -m pytest tests/test_dist.py tests/test_abi.py -x -q -m gpu -p no:cacheprovider ) 2>&1 | tail -8
