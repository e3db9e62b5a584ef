"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / SURVEY.md 8(e).

One MSM is split by point range; each rank produces a 128-byte partial sum for
its range, the ranks all-gather the partials, and the product's
bpg_msm_combine_partials() (host code of libbpg.so, no GPU needed) adds and
compresses them.  On the GPU box the per-rank partial comes from the HIP
kernels; here the oracle stands in for the per-rank MSM so that the exchange and
the combine can be exercised without a GPU.  The result must equal the unsharded
MSM's encoding.
"""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_per_rank, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bulletproofs_gadgets_amd import _native, synth
    from oracle import binding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = binding.load()
    # the same per-rank inputs bench.py derives
    u = synth.uniform_bytes(n_per_rank, tag=b"P/%d" % rank)
    sc = synth.scalars(n_per_rank, tag=b"s/%d" % rank)
    arr, n = orc.from_uniform(u)
    part = orc.msm_partial(sc, arr, n)
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    allp = torch.empty(128 * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(allp, mine)
    combined = _native.combine_partials(allp.numpy().tobytes())
    # every rank checks against the unsharded MSM over all ranges
    full_u = b"".join(synth.uniform_bytes(n_per_rank, tag=b"P/%d" % r) for r in range(world))
    full_s = b"".join(synth.scalars(n_per_rank, tag=b"s/%d" % r) for r in range(world))
    farr, fn = orc.from_uniform(full_u)
    want = orc.msm(full_s, farr, fn)
    q.put((rank, combined == want, combined.hex()))
    dist.barrier()
    dist.destroy_process_group()


def _worker_exchange(rank, world, port, q):
    """the all-gather the sharded prover is given (r1cs.make_all_gather): rank-major layout, several targets,
    and the per-target combine the C++ side performs on it (finish_partials), restated with the product's
    host combine"""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bulletproofs_gadgets_amd import _native, synth
    from bulletproofs_gadgets_amd import r1cs as g
    from oracle import binding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = binding.load()
    all_gather = g.make_all_gather(dist, None)
    T, n = 3, 40                                  # three targets over this rank's 40 generators
    u = synth.uniform_bytes(n, tag=b"X/%d" % rank)
    arr, _ = orc.from_uniform(u)
    rows = [synth.scalars(n, tag=b"x/%d/%d" % (rank, t)) for t in range(T)]
    mine = b"".join(orc.msm_partial(rows[t], arr, n) for t in range(T))          # T x 128 B, as msm_wait(partial)
    allp = all_gather(mine)
    ok = len(allp) == world * len(mine) and allp[len(mine) * rank:len(mine) * (rank + 1)] == mine
    outs = []
    for t in range(T):
        one = b"".join(allp[r * len(mine) + 128 * t:r * len(mine) + 128 * (t + 1)] for r in range(world))
        outs.append(_native.combine_partials(one))
    # reference: the unsharded MSM per target over all ranks' generators
    full_u = b"".join(synth.uniform_bytes(n, tag=b"X/%d" % r) for r in range(world))
    farr, fn = orc.from_uniform(full_u)
    for t in range(T):
        full_s = b"".join(synth.scalars(n, tag=b"x/%d/%d" % (r, t)) for r in range(world))
        ok = ok and outs[t] == orc.msm(full_s, farr, fn)
    q.put((rank, ok, b"".join(outs).hex()))
    dist.barrier()
    dist.destroy_process_group()


def test_partial_sum_exchange_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1


def test_point_range_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 300, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1


def test_combine_rejects_garbage():
    sys.path.insert(0, ROOT)
    from bulletproofs_gadgets_amd import _native
    with pytest.raises(_native.BpgError) as e:
        _native.combine_partials(bytes(range(128)))
    assert e.value.code == _native.EFORMAT
    # identity partials combine to the identity encoding
    ident = (0).to_bytes(32, "little") + (1).to_bytes(32, "little") + (1).to_bytes(32, "little") + (0).to_bytes(32, "little")
    assert _native.combine_partials(ident * 3) == bytes(32)
