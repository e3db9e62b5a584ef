"""Write the scalar_mul statement of bench.py's secondary metric for tools/native/prove_driver:
    python tools/dump_statement.py NBITS OUT"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_gadgets_amd import synth  # noqa: E402
from bulletproofs_gadgets_amd.r1cs import _xyzt  # noqa: E402

nbits = int(sys.argv[1])
base, bits, expected = synth.scalar_mul_statement(nbits)
with open(sys.argv[2], "wb") as f:
    f.write(_xyzt(base))
    f.write(_xyzt(expected))
    f.write(struct.pack("<I", nbits))
    f.write(b"".join(int(b).to_bytes(32, "little") for b in bits))
