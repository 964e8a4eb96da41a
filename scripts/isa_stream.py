#!/usr/bin/env python3
"""Compressed instruction stream of one kernel's innermost loop from a --save-temps .s file (M = MFMA, R = LDS read, W = LDS write,
G = LDS-DMA, L = global load, [..] = s_waitcnt, |BAR| = s_barrier, v / s = other vector / scalar).  usage: isa_stream.py FILE.s KERNEL_REGEX"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = re.compile(r'^_Z\S*' + sys.argv[2])
for start in [i for i, l in enumerate(lines) if pat.match(l) and l.rstrip().endswith(':') or (pat.match(l) and ':' in l)]:
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('; TotalNumVgprs'))
    body = [l.strip() for l in lines[start:end]]
    hdr = [i for i, l in enumerate(body) if 'Loop Header' in l]
    if not hdr:
        continue
    li = hdr[-1]
    le = [i for i in range(li, len(body)) if body[i].startswith('s_cbranch')][0]
    out = []
    for l in body[li:le + 1]:
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'): out.append('M')
        elif op == 's_waitcnt': out.append('[' + l.split(None, 1)[1].split(';')[0].strip() + ']')
        elif op.startswith('ds_read'): out.append('R')
        elif op.startswith('ds_write'): out.append('W')
        elif op.startswith('global_load_lds'): out.append('G')
        elif op.startswith('global_load'): out.append('L')
        elif op == 's_barrier': out.append('|BAR|')
        elif op.startswith('s_'): out.append('s')
        else: out.append('v')
    print(lines[start][:100])
    print([x for x in lines[end - 3:end + 1] if 'Vgprs' in x])
    print(''.join(out))
