#!/usr/bin/env python3
"""Static instruction count per '##PHASE' marker in a kernel's assembly (analysis helper)."""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
phase = 'prologue'
counts, order = {}, []
for l in lines:
    m = re.search(r'##PHASE (\w+)', l)
    if m:
        phase = m.group(1)
        if phase not in counts:
            order.append(phase)
        continue
    if re.match(r'^\s+[a-z_0-9]+(\s|$)', l) and not l.strip().startswith((';', '.')):
        op = l.split()[0]
        d = counts.setdefault(phase, {})
        kind = 'S' if op.startswith('s_') else ('LDS' if op.startswith('ds_') else ('VMEM' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'V'))
        d[kind] = d.get(kind, 0) + 1
        if op in ('v_readlane_b32', 'v_writelane_b32'):
            d['lane'] = d.get('lane', 0) + 1
for p in ['prologue'] + order:
    print(f'{p:20s}', counts.get(p))
