#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for NP in 0 1; do for S in 1 32; do
if [ $NP = 1 ]; then export SPEEXHIP_NO_PAD=1; else unset SPEEXHIP_NO_PAD; fi
python bench.py --config cfg4 --streams $S --steps 30 --warmup 3 --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg4 nopad=$NP S=$S launch_us', d['roofline']['launch_us'], 'value', d['value'], 'hbm', d['roofline']['frac'], 'valu', d['valu']['frac'], 'fast_path', d['config']['fast_path'], d.get('parity'))"
done; done
unset SPEEXHIP_NO_PAD
for S in 1 32; do python bench.py --streams $S --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg2 S=$S launch_us', d['roofline']['launch_us'], 'value', d['value'])"; done
