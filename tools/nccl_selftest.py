"""The torch.distributed calls bench.py makes on the N > 1 path, over the nccl (= RCCL) backend.

  python3 tools/nccl_selftest.py                      world size 1: argument types, dtypes and devices are accepted
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
      tools/nccl_selftest.py                          world size 2 with BOTH ranks on GPU 0 (a one-GPU box): RCCL may refuse
                                                      two ranks on one device; the outcome is printed as one JSON line

With world size > 1 it also times the per-MSM exchange (a 128-byte all-gather, host bytes in and out, as
bulletproofs_gadgets_amd.r1cs.make_all_gather does it) and prints the latency.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    rec = {"world": world, "device_of_this_rank": dev, "gpus_visible": torch.cuda.device_count()}
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        gather_in = torch.empty(128, dtype=torch.uint8, device="cuda")
        gather_out = torch.empty(128 * world, dtype=torch.uint8, device="cuda")
        part = bytes((rank * 7 + i) & 255 for i in range(128))
        gather_in.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8))
        dist.all_gather_into_tensor(gather_out, gather_in)
        got = gather_out.cpu().numpy().tobytes()
        for r in range(world):
            assert got[128 * r:128 * r + 128] == bytes((r * 7 + i) & 255 for i in range(128))
        t = torch.tensor([1.5 + rank, 2.5, 3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.tolist() == [1.5 + world - 1, 2.5, 3.5]
        dist.barrier()
        torch.cuda.synchronize()
        from bulletproofs_gadgets_amd import r1cs as g
        ag = g.make_all_gather(dist, "cuda")
        assert ag(part)[128 * rank:128 * rank + 128] == part
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            ag(part)
            lat.append((time.perf_counter() - t0) * 1e6)
        lat.sort()
        rec.update({"ok": True, "exchange_us_median": round(lat[100], 1), "exchange_us_p90": round(lat[180], 1),
                    "what": "128-byte all-gather over nccl, host bytes -> device tensor -> all_gather_into_tensor -> host bytes"})
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001  (the outcome is the record)
        rec.update({"ok": False, "error": repr(e)[:400]})
    if rank == 0:
        print(json.dumps(rec), flush=True)
    print("nccl selftest %s (rank %d)" % ("ok" if rec.get("ok") else "FAILED", rank))
    return 0 if rec.get("ok") else 1


if __name__ == "__main__":
    sys.exit(main())
