#!/usr/bin/env python3
"""Per-kernel ISA statistics of a hipcc -S dump: basic blocks, the largest block's (= hot loop's) instruction mix and waits.
usage: tools/isa_stats.py file.s [kernel-name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(';')]
    blocks, cur, lab = [], [], 'entry'
    for l in lines:
        if re.match(r'\.LBB\d+_\d+:', l):
            blocks.append((lab, cur)); cur = []; lab = l
        elif not l.startswith('.'):
            cur.append(l.split(';')[0].strip())
    blocks.append((lab, cur))
    print(name)
    print('  blocks:', [(b[0], len(b[1])) for b in blocks if len(b[1]) > 20])
    big = max(blocks, key=lambda b: len(b[1]))
    cnt = collections.Counter(l.split()[0] for l in big[1])
    print('  hot block', big[0], len(big[1]), 'instr; VALU', sum(v for k, v in cnt.items() if k.startswith('v_')))
    print('   ', {k: v for k, v in sorted(cnt.items()) if not k.startswith('v_')})
    print('    valu mix:', {k: v for k, v in cnt.most_common() if k.startswith('v_')})
    print('    waits:', [l for l in big[1] if l.startswith('s_waitcnt')])
