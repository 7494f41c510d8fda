#!/usr/bin/env python3
"""Static instruction mix of one kernel of the built HIP library, split at the phase-timer stamps (s_memtime) the kernels carry.
usage: isa_phase_stats.py <objdump -d of the gfx950 code object> <substring of the kernel symbol> [--all]
Prints, per segment between two stamps: VALU (f64 fma/mul/add separately), MFMA, LDS, SALU, s_waitcnt, global loads / stores."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    inside = False
    seg = []
    segs = []
    for line in open(path):
        if line.startswith("0") and "<" in line and line.rstrip().endswith(">:"):
            if inside:
                break
            inside = key in line and ("ELi0EEE" in line or "--all" in sys.argv)
            continue
        if not inside:
            continue
        m = re.match(r"\s+(\S+)\s", line)
        if not m:
            continue
        op = m.group(1)
        if op == "s_memtime":
            segs.append(seg)
            seg = []
        else:
            seg.append(op)
    segs.append(seg)

    def cls(op):
        if "mfma" in op:
            return "mfma"
        if op.startswith("v_") and op.endswith("_f64") or "_f64_" in op:
            if "fma" in op:
                return "fma64"
            if "mul" in op or "add" in op:
                return "muladd64"
            return "other64"
        if op.startswith("v_"):
            return "valu"
        if op.startswith("ds_"):
            return "lds"
        if op == "s_waitcnt":
            return "wait"
        if op.startswith("s_"):
            return "salu"
        if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"):
            return "gload"
        if op.startswith("global_store") or op.startswith("scratch_store"):
            return "gstore"
        return "other"

    keys = ["fma64", "muladd64", "other64", "valu", "mfma", "lds", "salu", "wait", "gload", "gstore", "other"]
    print("seg   total " + " ".join("%8s" % k for k in keys))
    tot = dict.fromkeys(keys, 0)
    for i, s in enumerate(segs):
        c = dict.fromkeys(keys, 0)
        for op in s:
            c[cls(op)] += 1
        for k in keys:
            tot[k] += c[k]
        print("%3d %7d " % (i, len(s)) + " ".join("%8d" % c[k] for k in keys))
    print("all %7d " % sum(len(s) for s in segs) + " ".join("%8d" % tot[k] for k in keys))


if __name__ == "__main__":
    main()
