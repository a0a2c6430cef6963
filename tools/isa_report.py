#!/usr/bin/env python3
"""Static instruction mix and resources of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
vector / scalar / LDS / memory instruction counts, v_readlane / v_writelane (SGPR spills into VGPR lanes show up here),
registers, scratch, occupancy.   python tools/isa_report.py file.s [name filter]"""
import collections
import re
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name in re.findall(r'^(_Z\w+):', t, re.M):
    if flt not in name:
        continue
    b = t[t.index(name + ':'):]
    if 's_endpgm' not in b:
        continue
    meta = b[b.index('s_endpgm'):][:6000]
    b = b[:b.index('s_endpgm')]
    c = collections.Counter()
    for l in b.split('\n'):
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        op = s.split()[0]
        if op.endswith(':'):
            continue
        k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'other'
        c[k] += 1
        if op in ('v_readlane_b32', 'v_writelane_b32'):
            c['lane_rw'] += 1
        if op == 's_waitcnt':
            c['waitcnt'] += 1

    def g(k):
        m = re.search(r'; ' + k + r': (\d+)', meta)
        return m.group(1) if m else '?'
    print(name[:70], dict(c), 'vgpr', g('NumVgprs'), 'sgpr', g('NumSgprs'), 'scratch', g('ScratchSize'), 'occupancy', g('Occupancy'))
