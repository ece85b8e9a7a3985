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


def _launch(mode, timeout):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py'), mode]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1')
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    return text


def test_sharding_host_logic_two_ranks_cpu():
    text = _launch('cpu', 300)
    assert 'rank 0 cpu ok' in text and 'rank 1 cpu ok' in text


@pytest.mark.gpu
def test_sharded_hot_path_two_ranks_one_gpu():
    text = _launch('gpu', 900)
    assert 'rank 0 gpu ok' in text and 'rank 1 gpu ok' in text
