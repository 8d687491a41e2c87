#!/usr/bin/env python3
"""Per-loop instruction mix of one kernel in a gfx950 assembly listing.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S x.hip -o x.s
       python tools/asm_loops.py x.s <substring of the mangled kernel name>
"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*%s\S*):\s*(;.*)?\n' % re.escape(key), s, re.M)
if not m:
    sys.exit('no kernel matching ' + key)
print(m.group(1))
body = s[m.end():s.index('s_endpgm', m.end())]
lines = body.split('\n')
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = n
for n, l in enumerate(lines):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        a = labels[mm.group(1)]
        seg = lines[a:n]
        c = lambda k: sum(1 for x in seg if k in x)
        print(mm.group(1), 'len', n - a, 'pk_fma', c('v_pk_fma'), 's_load', c('s_load'), 'ds_read', c('ds_read'),
              'readlane', c('v_readlane'), 'writelane', c('v_writelane'), 'waitcnt', c('s_waitcnt'), 's_mov',
              c('s_mov'), 'v_mov', c('v_mov'))
