"""CPU: pin the oracle (oracle/*.c) to the golden vectors (libsodium / RFC 9496 /
hashlib / Merlin KAT) before anything trusts it.  No GPU."""
import ctypes
import hashlib
import json
import os

import pytest

L = 2**252 + 27742317777372353535851937790883648493
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


def _buf(n=32):
    return ctypes.create_string_buffer(n)


def test_basepoint_multiples_rfc9496(oracle, golden):
    # RFC 9496 A.1: multiples 0..15 of the generator
    assert golden["basepoint_multiples"][1] == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
    B = oracle.ge_array(1)
    oracle.lib.ge_basepoint(B)
    for k, want in enumerate(golden["basepoint_multiples"]):
        sc = k.to_bytes(32, "little")
        for sched in range(5):
            assert oracle.msm(sc, B, 1, sched).hex() == want


def test_from_uniform_bytes(oracle, golden):
    for v in golden["from_hash"]:
        arr, n = oracle.from_uniform(bytes.fromhex(v["in"]))
        assert oracle.encode(arr, 1).hex() == v["out"]


def test_decode_encode_roundtrip_and_invalid(oracle, golden):
    for v in golden["from_hash"]:
        arr, n = oracle.decode(bytes.fromhex(v["out"]))
        assert arr is not None and oracle.encode(arr, 1).hex() == v["out"]
    for bad in golden["invalid_encodings"]:
        arr, rc = oracle.decode(bytes.fromhex(bad))
        assert arr is None
    arr, n = oracle.decode(bytes(32))   # identity is valid
    assert arr is not None


def test_pedersen_and_bulletproof_gens(oracle, golden):
    B, Bb = oracle.pedersen()
    assert oracle.encode(B, 1).hex() == golden["basepoint"]
    assert oracle.encode(Bb, 1).hex() == golden["B_blinding"]
    G, H = oracle.gens(8)
    assert [oracle.encode(G, 8)[32 * i:32 * i + 32].hex() for i in range(8)] == golden["gens_G"]
    assert [oracle.encode(H, 8)[32 * i:32 * i + 32].hex() for i in range(8)] == golden["gens_H"]


def _msm_inputs(seed, n):
    pu = hashlib.shake_256(b"golden-msm" + seed + b"P").digest(64 * n)
    su = hashlib.shake_256(b"golden-msm" + seed + b"s").digest(64 * n)
    sc = b"".join((int.from_bytes(su[64 * i:64 * i + 64], "little") % L).to_bytes(32, "little") for i in range(n))
    return pu, sc


def test_msm_all_schedules_match_golden(oracle, golden):
    for v in golden["msm"]:
        pu, sc = _msm_inputs(v["seed"].encode(), v["n"])
        arr, n = oracle.from_uniform(pu)
        scheds = [0, 1, 2, 3] + ([4] if n <= 17 else [])
        for s in scheds:
            assert oracle.msm(sc, arr, n, s).hex() == v["out"], (v["n"], s)


def test_msm_edge_scalars(oracle, golden):
    e = golden["msm_edge"]
    pts = b"".join(bytes.fromhex(p) for p in e["points"])
    sc = b"".join(bytes.fromhex(s) for s in e["scalars"])
    for s in range(5):
        assert oracle.msm_compressed(sc, pts, s).hex() == e["out"]


def test_reduce_wide(oracle, golden):
    for v in golden["reduce_wide"]:
        r = _buf()
        oracle.lib.sc_frombytes_wide(r, bytes.fromhex(v["in"]))
        assert r.raw.hex() == v["out"]


def test_merlin_kat(oracle, golden):
    k = golden["merlin_kat"]
    t = _buf(256)
    oracle.lib.merlin_init(t, k["protocol"].encode(), len(k["protocol"]))
    oracle.lib.merlin_append_message(t, k["label"].encode(), k["data"].encode(), len(k["data"]))
    out = _buf()
    oracle.lib.merlin_challenge_bytes(t, k["challenge_label"].encode(), out, 32)
    assert out.raw.hex() == k["out"]


def test_keccak_matches_hashlib(oracle):
    for n in (0, 1, 71, 72, 135, 136, 137, 500):
        msg = bytes(range(256)) * 2
        msg = msg[:n]
        k = _buf(256)
        oracle.lib.shake256_init(k)
        oracle.lib.keccak_absorb(k, msg, len(msg))
        o = _buf(300)
        oracle.lib.shake256_squeeze(k, o, 300)
        assert o.raw == hashlib.shake_256(msg).digest(300)
        k = _buf(256)
        oracle.lib.sha3_512_init(k)
        oracle.lib.keccak_absorb(k, msg, len(msg))
        o = _buf(64)
        oracle.lib.sha3_512_final(k, o)
        assert o.raw == hashlib.sha3_512(msg).digest()


def test_scalar_arithmetic(oracle):
    import random
    rnd = random.Random(5)
    vals = [0, 1, 2, L - 1, L - 2] + [rnd.randrange(L) for _ in range(200)]
    r = _buf()
    for a in vals[:40]:
        for b in vals[:40]:
            A, B = a.to_bytes(32, "little"), b.to_bytes(32, "little")
            oracle.lib.sc_mul(r, A, B)
            assert int.from_bytes(r.raw, "little") == a * b % L
            oracle.lib.sc_add(r, A, B)
            assert int.from_bytes(r.raw, "little") == (a + b) % L
            oracle.lib.sc_sub(r, A, B)
            assert int.from_bytes(r.raw, "little") == (a - b) % L
    for a in vals[1:30]:
        oracle.lib.sc_invert(r, a.to_bytes(32, "little"))
        assert int.from_bytes(r.raw, "little") == pow(a, -1, L)
