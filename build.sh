#!/usr/bin/env bash
# Builds, in-tree:
#   bulletproofs_gadgets_amd/libbpg.so     the product (HIP kernels + C ABI), gfx950 only
#   oracle/liboracle.so                    the CPU checker (test infrastructure)
#   tests/native/libhost_hooks.so          host instantiation of the arithmetic headers (test only)
# hipcc cross-compiles for gfx950 without a GPU present.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=gfx950
SRC=bulletproofs_gadgets_amd/csrc
OUT=bulletproofs_gadgets_amd
mkdir -p build
HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=${ARCH} -Wall -Wno-unused-function -Iinclude"

objs=()
pids=()
for f in ${SRC}/*.hip; do
    o=build/$(basename "${f%.hip}").o
    # headers and the generated instruction stream (madd_gfx950.inc, included from hot25519.h) are dependencies of every object
    if [[ ! -f $o || $f -nt $o || -n $(find ${SRC} include \( -name '*.h' -o -name '*.inc' \) -newer "$o" -print -quit) ]]; then
        echo "hipcc $f"
        ( $HIPCC $HIPFLAGS -c "$f" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
        pids+=($!)
    fi
    objs+=("$o")
done
for p in ${pids[@]+"${pids[@]}"}; do wait "$p"; done      # a failed compile stops the build (set -e): never link a stale object
$HIPCC -shared -fPIC --offload-arch=${ARCH} -o ${OUT}/libbpg.so "${objs[@]}" -Wl,-rpath,/opt/rocm/lib -lpthread
echo "built ${OUT}/libbpg.so"

make -s -C oracle
echo "built oracle/liboracle.so"

$HIPCC -O2 -std=c++17 -fPIC --offload-arch=${ARCH} -shared -o tests/native/libhost_hooks.so tests/native/host_hooks.hip -Wl,-rpath,/opt/rocm/lib
echo "built tests/native/libhost_hooks.so"

# measurement tools (tools/*.hip): built on request, never committed
if [[ "${BPG_BUILD_TOOLS:-0}" == "1" ]]; then
    $HIPCC -O3 --offload-arch=${ARCH} tools/irate.hip -o tools/irate
    $HIPCC -O3 --offload-arch=${ARCH} tools/ubench.hip -o tools/ubench
    g++ -O2 -std=c++17 -Iinclude tools/native/prove_driver.cpp -o tools/native/prove_driver -Lbulletproofs_gadgets_amd -lbpg \
        -Wl,-rpath,'$ORIGIN/../../bulletproofs_gadgets_amd' -Wl,-rpath,/opt/rocm/lib -lpthread
    echo "built tools/irate tools/ubench tools/native/prove_driver"
fi
