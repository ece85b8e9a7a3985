"""world_size-2 coverage of the row-sharded multi-GPU path (gloo): host logic on CPU, full path on a GPU box."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_group(cmd, env, timeout):
    """Run `cmd` in its own session; on timeout kill the WHOLE process group (torchrun does not forward a SIGKILL it
    never sees, and orphaned ranks would keep the GPU while a second attempt starts).  -> (returncode | None, output)"""
    import signal
    import tempfile
    with tempfile.TemporaryFile() as log:
        p = subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            rc = None
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
        log.seek(0)
        return rc, log.read().decode(errors='replace')


def _launch(mode, timeout, world=2, **extra_env):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py'), mode]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1', **extra_env)
    # a healthy run takes 10 - 60 s.  One rendezvous that never completed (once in some thirty runs on the pool's boxes, the
    # same tree passing before and after) must not fail the suite: a launch in which NO rank got past init_process_group
    # is retried once on a fresh port.  A hang after the rendezvous is a deadlock of the path under test and fails with the log.
    rc, text = _run_group(cmd, env, min(timeout, 400))
    if rc is None and 'rendezvous done' not in text:
        cmd[cmd.index('--master-port') + 1] = str(_free_port())
        rc, text = _run_group(cmd, env, min(timeout, 400))
    assert rc is not None, 'timed out after the rendezvous (a collective never returned?):\n' + text[-4000:]
    assert rc == 0, text[-4000:]
    return text


def test_sharding_host_logic_two_ranks_cpu():
    text = _launch('cpu', 300)
    assert 'rank 0 cpu ok' in text and 'rank 1 cpu ok' in text


@pytest.mark.gpu
def test_sharded_hot_path_two_ranks_one_gpu():
    text = _launch('gpu', 900)
    assert 'rank 0 gpu ok' in text and 'rank 1 gpu ok' in text


@pytest.mark.gpu
def test_sharded_hot_path_three_ranks_one_gpu():
    """An odd number of ranks: row blocks of unequal size, the last one ragged (gloo, all ranks on cuda:0)."""
    text = _launch('gpu', 1200, world=3, B3C_DIST_WORKER_MAPS='1')       # the first map only: the three-map run is the two-rank test
    assert all('rank %d gpu ok' % r in text for r in range(3))


@pytest.mark.gpu
def test_sharded_aggregation_beyond_the_bin_table_two_ranks_one_gpu():
    """level 0 of a row-sharded run ending with > 4096 communities (C5's shape), and the pruned scan on row blocks"""
    text = _launch('gpu-many', 900)
    assert 'rank 0 gpu-many ok' in text and 'rank 1 gpu-many ok' in text


@pytest.mark.gpu
def test_aggregation_by_owner_three_ranks_one_gpu():
    """the same maps with three ranks: unequal owner ranges of the new communities, three-way all-to-all of the reduced runs, row-block
    levels after the first (round 6)"""
    text = _launch('gpu-many', 1200, world=3)
    assert all('rank %d gpu-many ok' % r in text for r in range(3))


@pytest.mark.gpu
def test_sharded_hot_path_beyond_a_million_contigs_two_ranks_one_gpu():
    """the sharded path on a C5-shaped map of 1.2M contigs / 4.8e7 edges: balanced row blocks, level 0 beyond the bin table"""
    text = _launch('gpu-large', 900)
    assert 'rank 0 gpu-large ok' in text and 'rank 1 gpu-large ok' in text


@pytest.mark.gpu
def test_c4_workload_through_the_sharded_code_two_ranks_one_gpu():
    """BASELINE.json configs[3] (1M contigs / 100M edges on the contract generator, row-sharded) with two ranks on one GPU: identical
    to the single-rank run"""
    text = _launch('gpu-c4', 900)
    assert 'rank 0 gpu-c4 ok' in text and 'rank 1 gpu-c4 ok' in text


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 3])
def test_direct_transport_is_bitwise_the_collective_one(world):
    """the all-gathers as peer writes into IPC-mapped windows (two / three processes sharing one GPU) against the callback transport"""
    text = _launch('gpu-direct', 1200, world=world)
    assert all('rank %d gpu-direct ok' % r in text for r in range(world))


@pytest.mark.gpu
def test_c4_workload_through_the_direct_transport_two_ranks_one_gpu():
    text = _launch('gpu-direct', 1200, B3C_DIRECT_C4='1')
    assert 'rank 0 gpu-direct ok' in text and 'rank 1 gpu-direct ok' in text


@pytest.mark.gpu
def test_a_rank_with_an_empty_block_takes_part_two_ranks_one_gpu():
    text = _launch('gpu-empty', 900)
    assert 'rank 0 gpu-empty ok' in text and 'rank 1 gpu-empty ok' in text


def _bench_two_ranks(mode, expect_failure=False, **extra_env):
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--config', 'C2', '--mode', mode]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1', B3C_BENCH_BACKEND='gloo',
               B3C_BENCH_ONE_GPU='1', **extra_env)
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    if expect_failure:
        assert out.returncode != 0, 'a failed sharded leg must fail the run'
        assert len(lines) == 1
        return json.loads(lines[0])
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['unit'] == 'edges/s' and 'roofline' in d
    return d


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['auto', 'sharded', 'replicas'])
def test_bench_two_ranks_one_gpu(mode):
    """bench.py as the driver launches it for N = 2 (gloo + one shared GPU stand in for RCCL + two GPUs)."""
    d = _bench_two_ranks(mode)
    if mode == 'replicas':
        assert d['scaling'] == 'weak' and 'replicas x2' in d['config']['parallelism']
        return
    assert d['scaling'] == 'strong' and d['exchanges_per_step'] > 100
    assert d['raw_map_handed_over'] is True          # every step adopted its own resident copy of the raw map (a rank keeps its row block)
    assert 'one map over 2 GPUs by contig-row blocks' in d['config']['parallelism']
    if mode == 'auto':
        assert d['replicas']['value'] > 0 and d['replicas']['scaling'] == 'weak' and d['agrees_with_single_gpu'] is True


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_direct_transport():
    """the sharded leg of bench.py with the all-gathers as peer writes (B3C_DIST_TRANSPORT=direct): counted by kind, same agreement"""
    d = _bench_two_ranks('auto', B3C_DIST_TRANSPORT='direct')
    k = d['exchanges_per_step_by_kind']
    assert k['direct'] > 100 and k['direct'] + k['collective'] == d['exchanges_per_step'] and k['collective'] < 40, k
    assert 'direct peer writes' in d['transport'] and d['agrees_with_single_gpu'] is True


@pytest.mark.gpu
@pytest.mark.parametrize('how', ['exception', 'watchdog'])
def test_bench_fails_loudly_when_the_sharded_leg_fails(how):
    """a sharded leg that raises or hangs is a failed run (non-zero status, value null); the replica figure is kept beside it"""
    env = {'B3C_BENCH_FAIL_SHARDED': '1'} if how == 'exception' else {'B3C_BENCH_WATCHDOG': '0.01'}
    d = _bench_two_ranks('auto', expect_failure=True, **env)
    assert d['value'] is None and 'error' in d['sharded'] and d['replicas']['value'] > 0


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
    """The sharded leg with the real RCCL backend and ONE rank (B3C_DIST_FORCE keeps the sharded code path on for a
    world of one): every exchange goes C -> ctypes callback -> torch.distributed all_gather_into_tensor on libb3c's
    stream.  Exercises the RCCL plumbing a single GPU can exercise; the xGMI transport needs the driver's multi-GPU run."""
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--config', 'C2', '--no-cpu-baseline']
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1', B3C_BENCH_FORCE_SHARDED='1',
               B3C_DIST_FORCE='1')
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['scaling'] == 'strong' and d['exchanges_per_step'] > 100 and d['agrees_with_single_gpu'] is True
    assert 'sharded' not in d and d['replicas']['value'] > 0


def test_balanced_starts_host_logic():
    """dist.balanced_starts (row ranges 'balanced by nnz (prefix-sum over indptr)'): ascending cuts aligned to 64 rows
    that start at 0, end at n and leave no rank with more than its share plus one alignment block of rows -- on flat rows, on hub
    rows at the front (equal row counts would be 3x out of balance) and on degenerate inputs."""
    import numpy as np
    from bin3c_amd.dist import balanced_starts
    rs = np.random.RandomState(11)
    for world in (1, 2, 3, 8):
        for lens in (rs.randint(1, 400, 100000), np.concatenate((np.full(50, 500000), rs.randint(1, 50, 20000))),
                     np.zeros(1000, dtype=np.int64), np.ones(5, dtype=np.int64), np.zeros(0, dtype=np.int64)):
            lens = lens.astype(np.int64)
            st = balanced_starts(lens, world)
            n = lens.shape[0]
            assert st.shape == (world + 1,) and st[0] == 0 and st[-1] == n and np.all(np.diff(st) >= 0)
            assert np.all(st[1:-1] % 64 == 0) or n < 64 * world
            if lens.sum() > 0 and n >= 64 * world:
                per = np.array([lens[st[r]:st[r + 1]].sum() for r in range(world)])
                slack = max(int(lens[a:a + 64].sum()) for a in range(0, n, 64))       # one alignment block of rows
                assert per.max() <= lens.sum() / world + slack + 1, (world, per, slack)
