#!/usr/bin/env bash
# A/B of two builds of libbpg.so on the same GPU box: ab/libbpg_old.so vs the in-tree library.
# usage: tools/ab.sh <rounds> <command...>      (the command's stdout is appended to gpurun_out/ab.log)
set -euo pipefail
rounds=$1; shift
lib=bulletproofs_gadgets_amd/libbpg.so
cp $lib ab/libbpg_new.so
: > gpurun_out/ab.log
for r in $(seq "$rounds"); do
    for which in old new; do
        cp ab/libbpg_$which.so $lib
        echo "== $which $r" >> gpurun_out/ab.log
        timeout -k 10 300 "$@" >> gpurun_out/ab.log 2>&1
    done
done
cp ab/libbpg_new.so $lib
