"""The torch.distributed calls bench.py makes on the N > 1 path, over the nccl (= RCCL) backend with world size 1:
checks that the argument types, dtypes and devices are accepted on this image (one GPU is enough for that)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    gather_in = torch.empty(128, dtype=torch.uint8, device="cuda")
    gather_out = torch.empty(128, dtype=torch.uint8, device="cuda")
    part = bytes(range(128))
    gather_in.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8))
    dist.all_gather_into_tensor(gather_out, gather_in)
    assert gather_out.cpu().numpy().tobytes() == part
    t = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [1.5, 2.5, 3.5]
    dist.barrier()
    torch.cuda.synchronize()
    from bulletproofs_gadgets_amd import r1cs as g
    ag = g.make_all_gather(dist, "cuda")
    assert ag(part) == part
    dist.destroy_process_group()
    print("nccl selftest ok")


if __name__ == "__main__":
    main()
