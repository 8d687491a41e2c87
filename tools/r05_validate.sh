#!/bin/bash
# tools/r05_validate.sh -- the whole GPU suite, three fuzz passes (host calls; batched device calls; many-states calls), first call after warm-up
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05_validate; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 400 python tools/fuzz_gpu.py --many --seconds ${FUZZ_S:-200} --seed 51 > $O/fuzz_many.txt 2>&1; tail -3 $O/fuzz_many.txt
timeout 300 python tools/fuzz_gpu.py --seconds 120 --seed 52 > $O/fuzz_host.txt 2>&1; tail -2 $O/fuzz_host.txt
timeout 300 python tools/fuzz_gpu.py --batch --seconds 120 --seed 53 > $O/fuzz_batch.txt 2>&1; tail -2 $O/fuzz_batch.txt
FIRST_CALL_WARMUP=1 timeout 300 python tools/first_call.py > $O/first_call_warm.txt 2>&1; cat $O/first_call_warm.txt | cut -c1-300
