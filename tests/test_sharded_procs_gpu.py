"""GPU: the sharded proof as separate processes over torch.distributed (gloo, so that the two
ranks may share the box's single GPU; with one GPU per rank the same tool runs over RCCL).
Each rank proves sharded, then unsharded, and the bytes must agree on and across ranks.

The nccl tests arm themselves: they run whenever the box shows two or more GPUs (one rank per GPU over RCCL) and
skip, saying so, on a one-GPU box -- so the first multi-GPU machine that runs `pytest -m gpu` exercises the RCCL
exchange before any benchmark does."""
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


def _torchrun(nproc, script, *args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, script), *args]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_processes_full_size_proof():
    """BASELINE config 4 at its full size (38 836 bits = 2^20 multipliers) with the generators split over two
    PROCESSES: both ranks produce the unsharded prover's bytes (each rank also proves unsharded and compares), and
    the verifier accepts on every rank."""
    res = _torchrun(2, "tools/sharded_proof.py", "--nbits", "38836", "--backend", "gloo", "--repeat", "1", "--check", timeout=280)
    assert res["ranks"] == 2 and res["multipliers"] == 1 << 20
    assert res["all_ranks_agree_and_equal_unsharded"] is True
    assert res["proof_bytes"] == 1 + 32 * (13 + 2 * 20)       # version byte, 13 fixed elements, L and R of 20 rounds


def _gpus():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU in this process


def test_rccl_exchange_when_two_gpus():
    """one rank per GPU: the calls bench.py and make_all_gather make over nccl (= RCCL), and the 128-byte exchange"""
    if _gpus() < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device (profiles/r02_nccl_selftest.json); runs by itself on a multi-GPU box")
    res = _torchrun(2, "tools/nccl_selftest.py")
    assert res["ok"] is True and res["world"] == 2, res


def test_sharded_proof_over_rccl_when_two_gpus():
    if _gpus() < 2:
        pytest.skip("one GPU visible; runs by itself on a multi-GPU box")
    res = _torchrun(2, "tools/sharded_proof.py", "--nbits", "2427", "--backend", "nccl", "--repeat", "1", "--check")
    assert res["ranks"] == 2 and res["backend"] == "nccl"
    assert res["all_ranks_agree_and_equal_unsharded"] is True


def test_bench_two_ranks_over_rccl_when_two_gpus():
    """bench.py's N = 2 path as the driver launches it (point-range-sharded MSM, partial sums all-gathered over RCCL)"""
    if _gpus() < 2:
        pytest.skip("one GPU visible; runs by itself on a multi-GPU box")
    res = _torchrun(2, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-big-proof", "--no-batch", "--no-cpu-baseline",
                    "--no-extras", "--no-proofs")
    assert res["n_gpus"] == 2 and res["value"] > 0
