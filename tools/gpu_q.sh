#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for CFG in cfg3 f3; do for S in 1 32; do
python bench.py --config $CFG --streams $S --steps 30 --warmup 3 --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$CFG S=$S launch_us', d['roofline']['launch_us'], 'value', d['value'], 'hbm', d['roofline']['frac'], 'valu', d['valu']['frac'], 'fast_path', d['config']['fast_path'], d.get('parity'))"
done; done
