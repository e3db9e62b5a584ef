"""Attribute the frames of a glog stack trace (absolute addresses, no load bases recorded) to shared libraries.

A frame above the faulting one is a return address: the instruction before it is a call.  For a group of frames that
lie within one library's span, try every page-aligned load base and keep those at which EVERY frame of the group
follows a call instruction of that library; with three or more frames a match is unambiguous.  A matching base then
names the functions (nearest preceding dynamic symbol).

    python tools/symbolise_frames.py LIB ADDR [ADDR ...]        (addresses in hex)
"""
import bisect
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def return_addresses(lib):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", lib], capture_output=True, text=True).stdout
    rets, prev_call = set(), False
    for line in out.splitlines():
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)", line)
        if not m:
            continue
        addr = int(m.group(1), 16)
        if prev_call:
            rets.add(addr)
        prev_call = m.group(2).startswith("call")
    return rets


def symbols(lib):
    out = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True).stdout
    syms = []
    for line in out.splitlines():
        p = line.split(None, 2)
        if len(p) == 3 and p[1] in "TtWw":
            syms.append((int(p[0], 16), p[2]))
    syms.sort()
    return syms


def main():
    lib, addrs = sys.argv[1], [int(a, 16) for a in sys.argv[2:]]
    rets = return_addresses(lib)
    print("%s: %d call sites" % (lib, len(rets)))
    syms = symbols(lib)
    keys = [s[0] for s in syms]
    lo = min(addrs)
    hits = []
    for r in rets:                      # anchor the lowest frame on every call site with the same page offset
        if (r & 0xfff) != (lo & 0xfff):
            continue
        base = lo - r
        if all((a - base) in rets for a in addrs):
            hits.append(base)
    print("bases at which all %d frames follow a call: %s" % (len(addrs), [hex(b) for b in sorted(set(hits))]))
    for base in sorted(set(hits)):
        for a in addrs:
            off = a - base
            i = bisect.bisect_right(keys, off) - 1
            name = "%s+0x%x" % (syms[i][1], off - syms[i][0]) if i >= 0 else "?"
            print("  0x%x -> +0x%x  %s" % (a, off, name))


if __name__ == "__main__":
    main()
