"""CPU: the committed instruction stream of the hot kernel (csrc/madd_gfx950.inc) is what its generator emits now.
The generator also re-runs its emulator self-test (results against Python integers, wait-state rule) on the way."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_stream_equals_generator_output(tmp_path):
    out = tmp_path / "madd.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_madd_asm.py"), "--out", str(out)], check=True, cwd=ROOT,
                   stdout=subprocess.DEVNULL)
    committed = open(os.path.join(ROOT, "bulletproofs_gadgets_amd", "csrc", "madd_gfx950.inc")).read()
    assert out.read_text() == committed, "csrc/madd_gfx950.inc is stale: run python tools/gen_madd_asm.py and rebuild"
