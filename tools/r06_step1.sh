#!/bin/bash
# round 6, first lease: the new pinned-buffer tests, the whole GPU suite on the diag-split build, the pinned path bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "pinned or block_acquire" > $O/r06_pytest_pinned.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_pinned.txt
tail -5 $O/r06_pytest_pinned.txt
timeout 600 python tools/pinned_path_bench.py > $O/r06_pinned_path.json 2> $O/r06_pinned_path.err; echo "pinned bench rc=$?"
cat $O/r06_pinned_path.json | head -150
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r06_pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_gpu.txt
tail -5 $O/r06_pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_form.json 2> $O/r06_bench.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/r06_bench_driver_form.json'))
print(d['value'], d['ms_per_step'], d['roofline']['launch_us'], d['end_to_end'], d['end_to_end_streams'], d['pcie_peak'])"
