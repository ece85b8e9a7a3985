cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python -u -m pytest tests -q -m gpu -p no:cacheprovider ) 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
