#!/usr/bin/env python3
"""Memory instructions, waits, barriers and branches of a kernel, in program order, from a `hipcc -S` dump.
usage: tools/loop_mem.py file.s kernel-name-substring [start-label]   (numbers = instruction index from the start label / kernel entry)"""
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
start_label = sys.argv[3] if len(sys.argv) > 3 else None
for chunk in s.split("\n\t.globl")[1:]:
    name = chunk.split()[0]
    if pat not in name:
        continue
    body = chunk[chunk.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    print(name)
    n, on = 0, start_label is None
    for l in body:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".LBB"):
            if start_label and t.startswith(start_label + ":"):
                on = True
            if on:
                print(t.split(";")[0].strip(), ("; " + t.split(";", 1)[1].strip()) if ";" in t else "")
            continue
        if not on:
            continue
        n += 1
        op = t.split()[0]
        if op.startswith(("global_", "ds_", "s_waitcnt", "s_barrier", "s_cbranch", "buffer_", "scratch_", "s_branch", "s_endpgm")):
            print("   ", n, t[:110])
    print("total", n)
