#!/usr/bin/env python3
"""Hot-path instruction counts of the big loops in an AMDGPU assembly listing (tools/isa_only.sh -> /tmp/isa/fast.s).

The kernels carry way-points (SY_HOT(tag) -> "; SYHOT tag" comments in the SY_ISA_ONLY build).  Per role (h_ = helper
wave, m_ = move wave) the tool finds the cheapest cycle through the role's way-points in order, where an exec-mask
skip (s_cbranch_execz) is never taken (lanes are active on the hot path) and a uniform branch (scc / vcc) goes the
cheaper way (restart, Philox refill, police-collision order and spin loops are the expensive sides).
Prints instruction classes per role and per segment; --dump writes the cycle's instructions to <out>.<role>.txt.
usage: python tools/isa_hot.py /tmp/isa/fast.s [--dump /tmp/isa/hot] [--min 200]
"""
import collections
import re
import sys

src = sys.argv[1]
dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
MIN = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 200
ins, labels = [], {}
for raw in open(src):
    l = raw.strip()
    if not l or l.startswith('.') and not l.endswith(':'):
        continue
    if l.endswith(':'):
        labels[l[:-1]] = len(ins)
        continue
    ins.append(l)          # (way-points "SYHOT tag" stay in the list as zero-cost pseudo instructions)


def cls(op):
    if op == 'SYHOT': return 'MARK'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'VMEM'
    if op.startswith('s_waitcnt'): return 'WAIT'
    if op.startswith(('s_load', 's_buffer', 's_memtime')): return 'SMEM'
    if op.startswith(('s_cbranch', 's_branch')): return 'BRANCH'
    if op.startswith('s_nop'): return 'NOP'
    if op.startswith('s_'): return 'SALU'
    return 'OTHER'


def target(i):
    m = re.match(r'(s_cbranch_\w+|s_branch)\s+(\S+)', ins[i])
    return (m.group(1), labels.get(m.group(2))) if m else (None, None)


ORDER = {'h': ['h_belrec', 'h_belstep', 'h_row'], 'm': ['m_moves', 'm_visits', 'm_maskcopy', 'm_eval', 'm_rewards', 'm_rew_fast']}


def shortest(src_i, goals):
    """Cheapest path from src_i to the nearest of goals (leaving src_i first, so src_i may be its own goal)."""
    import heapq
    dist, prev = {}, {}
    pq = []

    def succ(i):
        op, t = target(i)
        if op is None: return [i + 1]
        if op == 's_branch': return [t]
        if op == 's_cbranch_execz': return [i + 1]
        return [i + 1, t]
    w0 = 0 if cls(ins[src_i].split()[0]) in ('WAIT', 'MARK') else 1
    for c in succ(src_i):
        if c is not None and c < len(ins) and w0 < dist.get(c, 10 ** 9):
            dist[c] = w0; prev[c] = src_i; heapq.heappush(pq, (w0, c))
    goals = set(goals)
    while pq:
        d, i = heapq.heappop(pq)
        if d > dist.get(i, 10 ** 9): continue
        if i in goals:
            path = [i]
            while path[-1] != src_i or len(path) == 1:
                path.append(prev[path[-1]])
                if path[-1] == src_i: break
            return path[::-1]
        w = 0 if cls(ins[i].split()[0]) in ('WAIT', 'MARK') else 1
        for c in succ(i):
            if c is None or c >= len(ins): continue
            if d + w < dist.get(c, 10 ** 9):
                dist[c] = d + w; prev[c] = i; heapq.heappush(pq, (d + w, c))
    return None


marks = {}
for i, l in enumerate(ins):
    if l.startswith('SYHOT'): marks.setdefault(l.split()[1], []).append(i)
n = 0
for role, order in ORDER.items():
    tags = [t for t in order if t in marks]
    if not tags: continue
    # the hot cycle: first way-point -> ... -> last -> back to the first (every instance of the first tag tried)
    bestp = None
    for start in marks[tags[0]]:
        path, cur, ok = [start], start, True
        for t in tags[1:] + [tags[0]]:
            seg = shortest(cur, marks[t] if t != tags[0] else [start])
            if seg is None: ok = False; break
            path += seg[1:]; cur = seg[-1]
        if ok and (bestp is None or len(path) < len(bestp)): bestp = path
    if bestp is None:
        print('role %s: no cycle through %s' % (role, tags)); continue
    path = bestp[:-1]
    c = collections.Counter(cls(ins[i].split()[0]) for i in path)
    total = sum(v for k, v in c.items() if k not in ('WAIT', 'MARK'))
    seg_counts, cur, name = [], 0, None
    for i in path:
        if ins[i].startswith('SYHOT'):
            if name is not None: seg_counts.append((name, cur))
            cur = 0; name = ins[i].split()[1]
        elif cls(ins[i].split()[0]) != 'WAIT': cur += 1
    seg_counts.append((name, cur))
    print('role %s: hot cycle %d instructions  %s' % (role, total, dict(c)))
    print('        segments: ' + '  '.join('%s %d' % sc for sc in seg_counts))
    if dump:
        with open('%s.%s.txt' % (dump, role), 'w') as f:
            for i in path: f.write('%6d  %s\n' % (i, ins[i]))
