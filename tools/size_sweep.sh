#!/usr/bin/env bash
# proofs/s vs circuit size at 16 threads x 8 statements per call: work-bound (rate ~ 1/size) or latency-bound (flat)?
set -euo pipefail
for bits in 303 606 1213 2427; do
    python3 - "$bits" <<'PY'
import sys, runpy
import bench
bench.PROOF_BITS = int(sys.argv[1])
sys.argv = ["prove_many_timing.py", "8", "16"]
runpy.run_path("tools/prove_many_timing.py", run_name="__main__")
PY
done
