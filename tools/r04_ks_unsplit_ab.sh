#!/bin/bash
# tools/r04_ks_unsplit_ab.sh -- the sweep of tools/perf_sweep.sh with and without tap-range shares on unsplit launches
# (SPEEXHIP_KS_UNSPLIT=0 SPEEXHIP_TOUCH=0: the rules before; OFF= / ON= set other pairs of environments), same box; rows whose launch time moved by more than 3 %.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04${TAG:-}; mkdir -p $O
env ${OFF:-SPEEXHIP_KS_UNSPLIT=0 SPEEXHIP_TOUCH=0} tools/perf_sweep.sh > $O/sweep_ks_off.txt 2>&1
env ${ON:-X=1} tools/perf_sweep.sh > $O/sweep_ks_on.txt 2>&1
python3 - <<'PY'
import re
def rows(p):
    out = {}
    for l in open(p):
        m = re.match(r'([\d.]+) valu \|\s+([\d.]+) us .*\| (ch \d+ \d+ -> \d+) \| taps', l)
        if m: out[m.group(3)] = (float(m.group(2)), float(m.group(1)))
    return out
import os
O = 'gpurun_out/r04' + os.environ.get('TAG', '')
a, b = rows(O + '/sweep_ks_off.txt'), rows(O + '/sweep_ks_on.txt')
print('%d rows; moved by more than 3 %%:' % len(a))
for k in sorted(a, key=lambda k: b[k][0] / a[k][0]):
    r = b[k][0] / a[k][0]
    if abs(r - 1) > 0.03: print('  %-24s %8.1f -> %8.1f us  (%+5.1f %%)  valu %.3f -> %.3f' % (k, a[k][0], b[k][0], 100 * (r - 1), a[k][1], b[k][1]))
print('rows under 0.20 valu now:', sorted((v[1], k) for k, v in b.items() if v[1] < 0.20))
PY
