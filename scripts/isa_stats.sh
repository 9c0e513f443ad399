#!/bin/bash
# usage: scripts/isa_stats.sh <translation unit (e.g. kernels_fused_op9c)> <mangled-name substring>  -> instruction mix of that kernel's main loop
TU=$1; PAT=$2
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Imultigridcmt_amd/csrc $EXTRA \
   -S --cuda-device-only -o /tmp/isa/$TU.s multigridcmt_amd/csrc/$TU.hip 2>/dev/null || exit 1
python3 - "$TU" "$PAT" <<'PY'
import re, sys, collections
tu, pat = sys.argv[1], sys.argv[2]
txt = open('/tmp/isa/%s.s' % tu).read()
# split into functions
funcs = re.split(r'\n(?=_Z\w+:)', txt)
for fn in funcs:
    name = fn.split(':', 1)[0]
    if pat not in name or not name.startswith('_Z'):
        continue
    lines = fn.split('\n')
    # find the hottest loop: the largest backward branch range
    labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    meta = re.search(r'\.vgpr_count:\s+(\d+)', txt[txt.find('.name:           ' + name):])
    body = lines[best[0]:best[1]] if best else lines
    mix = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
            continue
        op = l.split()[0]
        key = ('fma64' if op.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64')) else
               'mov' if op.startswith(('v_mov', 'v_accvgpr', 'v_pk_mov')) else
               'cndmask' if op.startswith('v_cndmask') else
               'dpp/perm' if ('dpp' in l or 'permlane' in op) else
               'ds' if op.startswith('ds_') else
               'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
               'valu_other' if op.startswith('v_') else
               'waitcnt' if op.startswith('s_waitcnt') else
               'salu')
        mix[key] += 1
    print(name[:110])
    print('  vgprs', meta.group(1) if meta else '?', ' loop lines', len(body), dict(mix))
PY
