#!/usr/bin/env python3
"""Generates tests/golden/golden.json -- vectors that pin the oracle (and, through
it, the GPU path) to an INDEPENDENT ristretto255 implementation: libsodium 1.0.18
(/opt/conda/lib/libsodium.so.23, present only in the build container), plus
hashlib SHA3/SHAKE and Python integers.  The reference repository holds no golden
vectors of its own, and its Rust dependencies cannot be
built here, so this is the strongest pin available.

Run in the build container:  python tests/golden/gen_golden.py
"""
import ctypes
import hashlib
import json
import os

L = 2**252 + 27742317777372353535851937790883648493
S = ctypes.CDLL("/opt/conda/lib/libsodium.so.23")
assert S.sodium_init() >= 0


def buf(n=32):
    return ctypes.create_string_buffer(n)


def from_hash(h):
    r = buf()
    assert S.crypto_core_ristretto255_from_hash(r, h) == 0
    return r.raw


def smul(s, p):
    """s*P as ristretto bytes; identity handled (libsodium returns -1 for it)."""
    if s % L == 0:
        return bytes(32)
    r = buf()
    rc = S.crypto_scalarmult_ristretto255(r, (s % L).to_bytes(32, "little"), p)
    return r.raw if rc == 0 else bytes(32)


def add(p, q):
    r = buf()
    assert S.crypto_core_ristretto255_add(r, p, q) == 0
    return r.raw


def msm(scalars, points):
    acc = bytes(32)
    for s, p in zip(scalars, points):
        acc = add(acc, smul(s, p))
    return acc


def main():
    out = {"_source": "libsodium 1.0.18 ristretto255 + hashlib + Python ints; see gen_golden.py"}
    one = (1).to_bytes(32, "little")
    B = buf()
    S.crypto_scalarmult_ristretto255_base(B, one)
    B = B.raw
    out["basepoint"] = B.hex()
    out["basepoint_multiples"] = [smul(k, B).hex() for k in range(16)]
    out["from_hash"] = []
    for i in range(16):
        h = hashlib.sha512(b"bpg-golden-%d" % i).digest()
        out["from_hash"].append({"in": h.hex(), "out": from_hash(h).hex()})
    # PedersenGens::default().B_blinding = from_uniform_bytes(SHA3-512(compress(B)))
    out["B_blinding"] = from_hash(hashlib.sha3_512(B).digest()).hex()
    # BulletproofGens party-0 chains (label bytes recalled from bulletproofs generators.rs)
    for tag in (b"G", b"H"):
        stream = hashlib.shake_256(b"GeneratorsChain" + tag + (0).to_bytes(4, "little")).digest(64 * 8)
        out["gens_" + tag.decode()] = [from_hash(stream[64 * i:64 * i + 64]).hex() for i in range(8)]
    # MSM vectors: points from_hash(SHAKE256(seed|"P")), scalars reduce_wide(SHAKE256(seed|"s"))
    out["msm"] = []
    for seed in (b"a", b"b"):
        for n in (1, 2, 3, 17, 189, 190, 600, 1024):
            pu = hashlib.shake_256(b"golden-msm" + seed + b"P").digest(64 * n)
            su = hashlib.shake_256(b"golden-msm" + seed + b"s").digest(64 * n)
            pts = [from_hash(pu[64 * i:64 * i + 64]) for i in range(n)]
            scs = [int.from_bytes(su[64 * i:64 * i + 64], "little") % L for i in range(n)]
            out["msm"].append({"seed": seed.decode(), "n": n, "out": msm(scs, pts).hex()})
    # edge scalars on fixed points
    pts = [from_hash(hashlib.sha512(b"edge-%d" % i).digest()) for i in range(8)]
    scs = [0, 1, L - 1, 2, L - 2, 2**252, (L - 1) // 2, (L + 1) // 2]
    out["msm_edge"] = {"points": [p.hex() for p in pts], "scalars": [s.to_bytes(32, "little").hex() for s in scs],
                       "out": msm(scs, pts).hex()}
    # Scalar::from_bytes_mod_order_wide
    out["reduce_wide"] = []
    for i in range(8):
        w = hashlib.sha512(b"wide-%d" % i).digest()
        out["reduce_wide"].append({"in": w.hex(), "out": (int.from_bytes(w, "little") % L).to_bytes(32, "little").hex()})
    out["reduce_wide"].append({"in": (b"\xff" * 64).hex(), "out": ((2**512 - 1) % L).to_bytes(32, "little").hex()})
    # Merlin cross-language known-answer test (merlin crate test `equivalence_simple`, recalled)
    out["merlin_kat"] = {"protocol": "test protocol", "label": "some label", "data": "some data", "challenge_label": "challenge",
                         "out": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"}
    # invalid encodings (RFC 9496 A.3 classes): non-canonical, negative, non-square
    p = 2**255 - 19
    out["invalid_encodings"] = [x.to_bytes(32, "little").hex() for x in (p, p + 1, 1, 2**255, 2**256 - 1)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
