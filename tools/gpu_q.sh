#!/bin/bash
# quick: single-stream bench + HBM fetch counter
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -q -k "golden or batched or baseline" 2>&1 | tail -2
python bench.py --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('S=1 launch_us', d['roofline']['launch_us'], 'value', d['value'])"
cd /tmp && export TMPDIR=/tmp; rm -rf $O/pmcq
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --output-format csv -d $O/pmcq/$C -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > /dev/null 2>&1; done
python3 - <<'PY'
import csv, glob
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    v = [float(r['Counter_Value']) for f in glob.glob('/root/repo/gpurun_out/pmcq/%s/*/*counter_collection.csv' % c) for r in csv.DictReader(open(f)) if 'resample_' in r['Kernel_Name']]
    print(c, 'avg KiB per dispatch', sum(v) / len(v), 'n', len(v))
PY
