#!/usr/bin/env python3
"""Instruction-class shape of a kernel in a hipcc .s file: M = MFMA, v = VALU, d = LDS, g = global / buffer, w = s_waitcnt, B = barrier,
J = branch, s = other scalar; runs are written as <class><count>.  usage: isa_shape.py file.s <mangled-name-substring> [start [len]]"""
import re
import sys


def shape(path, sub):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    seq = []
    for l in lines[start + 1:end]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        seq.append("M" if op.startswith("v_mfma") else "v" if op.startswith("v_") else "d" if op.startswith("ds_") else
                   "g" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "w" if op.startswith("s_waitcnt") else
                   "B" if op.startswith("s_barrier") else "J" if op.startswith(("s_cbranch", "s_branch")) else "s")
    return "".join(seq)


if __name__ == "__main__":
    t = shape(sys.argv[1], sys.argv[2])
    print(len(t), "instructions;", {c: t.count(c) for c in "Mvdgw"})
    a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n = int(sys.argv[4]) if len(sys.argv) > 4 else len(t)
    print(re.sub(r"(.)\1*", lambda m: m.group(1) + (str(len(m.group(0))) if len(m.group(0)) > 1 else ""), t[a:a + n]))
