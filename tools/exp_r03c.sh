mkdir -p gpurun_out/r03c; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dist.py -m gpu -x -q -k "beyond or empty" > gpurun_out/r03c/pytest_dist.log 2>&1; tail -5 gpurun_out/r03c/pytest_dist.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03c/bench_$tag.json 2> gpurun_out/r03c/bench_$tag.err; echo "== $tag"; tail -1 gpurun_out/r03c/bench_$tag.json | python tools/bsum.py; }
run base B3C_X=1
run lvdebug B3C_LV_DEBUG=1
grep "level" gpurun_out/r03c/bench_lvdebug.err | tail -12
run incr2 B3C_LV_INCR_DIV=2
run incr1 B3C_LV_INCR_DIV=1
run order2 B3C_ORDER_ROUNDS=2
timeout 600 python tools/fuzz_ill.py 6000 1 > gpurun_out/r03c/fuzz_ill.log 2>&1; tail -8 gpurun_out/r03c/fuzz_ill.log
