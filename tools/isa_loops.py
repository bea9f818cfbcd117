#!/usr/bin/env python3
"""Per-loop instruction census of one kernel in a hipcc -S listing: VALU / DPP / LDS / scratch counts per loop body.
usage: isa_loops.py listing.s mangled_kernel_name_substring [...]"""
import re
import sys
from collections import Counter


def census(src, needle):
    m = re.search(r'^(\S*%s\S*): ' % re.escape(needle), src, re.M)
    if not m:
        print(needle, 'not found')
        return
    i = m.start()
    j = src.index('.end_amdhsa_kernel', i)
    b = src[i:j].split('\n')
    labels = {}
    for n, l in enumerate(b):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = n
    loops = []
    for n, l in enumerate(b):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            loops.append((labels[mm.group(1)], n))
    print(m.group(1)[-60:], 'lines', len(b))
    loops.sort()
    merged = []  # overlapping back edges (multi-entry loops, partial trips) count as one region
    for a, e in loops:
        if merged and a <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(e, merged[-1][1]))
        else:
            merged.append((a, e))
    for a, e in merged:
        seg = [l for l in b[a:e] if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
        ins = Counter(l.split()[0] for l in seg)
        g = lambda p: sum(v for k, v in ins.items() if k.startswith(p))
        dpp = sum(1 for l in seg if 'row_' in l or 'wave_sh' in l or 'quad_perm' in l)
        if len(seg) < 40:
            continue
        print('  loop @%d..%d instr %d valu %d (add_f32 %d mul_f32 %d fma %d dpp %d cvt %d) ds %d scratch %d vmem %d salu %d nop %d waitcnt %d' % (
            a, e, len(seg), g('v_'), g('v_add_f32'), g('v_mul_f32'), g('v_fma'), dpp, g('v_cvt'), g('ds_'), g('scratch_'),
            g('global_') + g('buffer_'), g('s_') - ins['s_nop'] - ins['s_waitcnt'], ins['s_nop'], ins['s_waitcnt']))


src = open(sys.argv[1]).read()
for needle in sys.argv[2:]:
    census(src, needle)
