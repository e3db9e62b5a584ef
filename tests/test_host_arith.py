"""CPU: the HOST instantiation of the product's host+device arithmetic headers
(fe25519.h / sc25519.h / ge25519.h, the very source the HIP kernels compile)
against Python big integers and the golden vectors.  No GPU, no oracle."""
import ctypes
import json
import os
import random

import pytest

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
HERE = os.path.dirname(os.path.abspath(__file__))


def B(x, n=32):
    return x.to_bytes(n, "little")


def I(b):
    return int.from_bytes(b.raw, "little")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


def test_field_ops_including_weakly_reduced_inputs(host_hooks):
    H = host_hooks
    rnd = random.Random(7)
    edge = [0, 1, 2, 19, 38, P - 1, P, P + 1, 2 * P, 2 * P + 37, 2**256 - 1, 2**256 - 38, 2**255, 2**255 - 1, 2**32 - 1, 2**224]
    vals = edge + [rnd.randrange(2**256) for _ in range(100)]
    o = ctypes.create_string_buffer(32)
    for a in vals[:40]:
        for b in vals[:40]:
            H.hh_fe_mul(o, B(a), B(b))
            assert I(o) == a * b % P
            H.hh_fe_add(o, B(a), B(b))
            assert I(o) == (a + b) % P
            H.hh_fe_sub(o, B(a), B(b))
            assert I(o) == (a - b) % P
    for a in vals:
        H.hh_fe_sq(o, B(a))
        assert I(o) == a * a % P
        H.hh_fe_mul_small(o, B(a), 0xffffffff)
        assert I(o) == a * 0xffffffff % P
    for a in vals[:30]:
        if a % P:
            H.hh_fe_invert(o, B(a))
            assert I(o) == pow(a, -1, P)


def test_scalar_ops(host_hooks, golden):
    H = host_hooks
    rnd = random.Random(9)
    vals = [0, 1, 2, L - 1, L - 2] + [rnd.randrange(L) for _ in range(100)]
    o = ctypes.create_string_buffer(32)
    for a in vals[:40]:
        for b in vals[:40]:
            H.hh_sc_mul(o, B(a), B(b))
            assert I(o) == a * b % L
            H.hh_sc_add(o, B(a), B(b))
            assert I(o) == (a + b) % L
            H.hh_sc_sub(o, B(a), B(b))
            assert I(o) == (a - b) % L
    for a in vals[1:25]:
        H.hh_sc_invert(o, B(a))
        assert I(o) == pow(a, -1, L)
    for x in [2**256 - 1, L, L + 1] + [rnd.randrange(2**256) for _ in range(100)]:
        H.hh_sc_reduce256(o, B(x))
        assert I(o) == x % L
    for v in golden["reduce_wide"]:
        H.hh_sc_reduce512(o, bytes.fromhex(v["in"]))
        assert o.raw.hex() == v["out"]


def test_scalar_mul_one_pass_reduction(host_hooks):
    """the host's a*b mod l folds the product above bit 252 three times (sc25519.h: sc_mul_host64): values at the
    edges of every fold -- products just below and above multiples of 2^252, operands up to 2^256 - 1 (callers pass
    canonical scalars, the arithmetic must hold for any 256-bit string) -- and 20 000 random pairs"""
    H = host_hooks
    rnd = random.Random(77)
    o = ctypes.create_string_buffer(32)
    C = L - 2**252
    edge = [0, 1, 2, L - 1, L - 2, L, L + 1, 2**252, 2**252 - 1, 2**252 + 1, C, C - 1, C + 1, 2**255 - 19, 2**256 - 1, 2**256 - 2, 2**128,
            2**128 - 1, 2**192, (L - 1) // 2, (L + 1) // 2, 2**253 - 1, 2**64 - 1, 2**64]
    for a in edge:
        for b in edge:
            H.hh_sc_mul(o, B(a), B(b))
            assert I(o) == a * b % L, (hex(a), hex(b))
    for _ in range(20000):
        bits_a, bits_b = rnd.choice((64, 125, 126, 200, 252, 253, 256)), rnd.choice((64, 125, 126, 200, 252, 253, 256))
        a, b = rnd.getrandbits(bits_a), rnd.getrandbits(bits_b)
        H.hh_sc_mul(o, B(a), B(b))
        assert I(o) == a * b % L, (hex(a), hex(b))


def test_ristretto_codec_and_group_ops(host_hooks, golden):
    H = host_hooks
    o = ctypes.create_string_buffer(32)
    for v in golden["from_hash"]:
        H.hh_from_uniform(o, bytes.fromhex(v["in"]))
        assert o.raw.hex() == v["out"]
        assert H.hh_ristretto_decode_encode(o, bytes.fromhex(v["out"])) == 0 and o.raw.hex() == v["out"]
    for bad in golden["invalid_encodings"]:
        assert H.hh_ristretto_decode_encode(o, bytes.fromhex(bad)) == -1
    # 2B + B = 3B, B + B = 2B, B - B = 0 through double / add / madd / msub
    m = [bytes.fromhex(x) for x in golden["basepoint_multiples"]]
    a, b, c = (ctypes.create_string_buffer(32) for _ in range(3))
    assert H.hh_group_ops(a, b, c, m[1], m[1]) == 0
    assert (a.raw, b.raw, c.raw) == (m[3], m[2], m[0])
    assert H.hh_group_ops(a, b, c, m[5], m[3]) == 0
    assert (a.raw, b.raw, c.raw) == (m[13], m[8], m[2])
