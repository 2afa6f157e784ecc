#!/usr/bin/env python3
"""Instruction histogram of an instruction-index range of a kernel in a `hipcc -S` dump (indices as printed by loop_mem.py / block list).
usage: tools/loop_hist.py file.s kernel-substring [first last]  — without a range: the list of blocks with their start index"""
import collections, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
rng = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
for chunk in s.split("\n\t.globl")[1:]:
    name = chunk.split()[0]
    if pat not in name:
        continue
    body = chunk[chunk.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    print(name)
    n = 0
    ops = collections.Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".LBB"):
            if rng is None:
                print(n, t[:100])
            continue
        n += 1
        if rng and rng[0] <= n <= rng[1]:
            ops[t.split()[0]] += 1
    if rng:
        tot = sum(ops.values())
        cat = collections.Counter()
        for k, v in ops.items():
            c = "VALU" if k.startswith("v_") else "SALU" if k.startswith("s_") else "LDS" if k.startswith("ds_") else "VMEM" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"
            cat[c] += v
        print("total", tot, dict(cat))
        print(", ".join(f"{k} {v}" for k, v in ops.most_common(40)))
    break
