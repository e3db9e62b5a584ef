"""GPU vs the committed golden vectors (libsodium / RFC 9496), through the C ABI.
Byte equality."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


def test_basepoint_multiples(gpu_ctx, golden):
    B = gpu_ctx.points_from_compressed(bytes.fromhex(golden["basepoint"]))
    for k, want in enumerate(golden["basepoint_multiples"]):
        assert gpu_ctx.msm(B, k.to_bytes(32, "little")).hex() == want


def test_from_uniform_and_decode(gpu_ctx, golden):
    u = b"".join(bytes.fromhex(v["in"]) for v in golden["from_hash"])
    want = b"".join(bytes.fromhex(v["out"]) for v in golden["from_hash"])
    assert gpu_ctx.points_from_uniform(u).compress() == want
    assert gpu_ctx.points_from_compressed(want).compress() == want


def test_generator_chain_prefix(gpu_ctx, golden):
    for tag in ("G", "H"):
        stream = hashlib.shake_256(b"GeneratorsChain" + tag.encode() + (0).to_bytes(4, "little")).digest(64 * 8)
        got = gpu_ctx.points_from_uniform(stream).compress()
        assert [got[32 * i:32 * i + 32].hex() for i in range(8)] == golden["gens_" + tag]


def test_msm_golden(gpu_ctx, golden):
    for v in golden["msm"]:
        seed, n = v["seed"].encode(), v["n"]
        pu = hashlib.shake_256(b"golden-msm" + seed + b"P").digest(64 * n)
        su = hashlib.shake_256(b"golden-msm" + seed + b"s").digest(64 * n)
        sc = b"".join((int.from_bytes(su[64 * i:64 * i + 64], "little") % L).to_bytes(32, "little") for i in range(n))
        pts = gpu_ctx.points_from_uniform(pu)
        assert gpu_ctx.msm(pts, sc).hex() == v["out"], (seed, n)


def test_msm_edge_golden(gpu_ctx, golden):
    e = golden["msm_edge"]
    pts = gpu_ctx.points_from_compressed(b"".join(bytes.fromhex(p) for p in e["points"]))
    sc = b"".join(bytes.fromhex(s) for s in e["scalars"])
    assert gpu_ctx.msm(pts, sc).hex() == e["out"]


def test_sharded_partials_combine(gpu_ctx, oracle):
    """point-range sharding on one GPU: two ranges -> partials -> combine == unsharded MSM."""
    from bulletproofs_gadgets_amd import _native, synth
    n = 2000
    u, sc = synth.uniform_bytes(n, tag=b"shard"), synth.scalars(n, tag=b"shard-s")
    pts = gpu_ctx.points_from_uniform(u)
    d = gpu_ctx.alloc(len(sc)).upload(sc)
    h = n // 2
    j1 = gpu_ctx.msm_submit(pts, d.ptr, h, segments=[(0, h)])
    j2 = gpu_ctx.msm_submit(pts, d.ptr + 32 * h, n - h, segments=[(h, n - h)])
    combined = _native.combine_partials(j1.wait_partial() + j2.wait_partial())
    assert combined == gpu_ctx.msm(pts, sc)
    arr, _ = oracle.from_uniform(u)
    assert combined == oracle.msm(sc, arr, n)
