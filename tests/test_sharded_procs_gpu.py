"""GPU: the sharded proof as separate processes over torch.distributed (gloo, so that the two
ranks may share the box's single GPU; with one GPU per rank the same tool runs over RCCL).
Each rank proves sharded, then unsharded, and the bytes must agree on and across ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_one_proof():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "sharded_proof.py"), "--nbits", "150",
           "--backend", "gloo", "--repeat", "1", "--check"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["ranks"] == 2 and res["multipliers"] == 27 * 150 + 4
    assert res["all_ranks_agree_and_equal_unsharded"] is True
