"""Opcode histogram of the main loop of one kernel in an ISA dump: isa_hist.py <file.s> <name substring> [n]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for fn in re.split(r'\n(?=_Z\w+:)', txt):
    name = fn.split(':', 1)[0]
    if pat not in name or not name.startswith('_Z'):
        continue
    lines = fn.split('\n')
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            sp = (labels[m.group(1)], i)
            if best is None or sp[1] - sp[0] > best[1] - best[0]:
                best = sp
    body = [l.strip() for l in lines[best[0]:best[1] + 1]]
    body = [l for l in body if l and not l.startswith(('.', ';')) and not l.endswith(':')]
    c = collections.Counter(l.split()[0] for l in body)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    s = sum(n for k, n in c.items() if k.startswith('s_') and not k.startswith('s_waitcnt'))
    print(name, '\n  loop instrs', len(body), 'VALU', v, 'SALU', s, 'ds', sum(n for k, n in c.items() if k.startswith('ds_')),
          'vmem', sum(n for k, n in c.items() if k.startswith(('global_', 'buffer_'))))
    print('  ' + '  '.join('%d %s' % (n, k) for k, n in c.most_common(top)))
