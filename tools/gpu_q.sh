#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for IO in int16 float; do for S in 1 32; do
python bench.py --io $IO --streams $S --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg2 io=$IO S=$S launch_us', d['roofline']['launch_us'], 'value', d['value'], 'hbm', d['roofline']['frac'], 'valu', d['valu']['frac'])"
done; done
python bench.py --steps 100 --warmup 10 | cut -c1-400
