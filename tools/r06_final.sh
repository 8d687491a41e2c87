#!/bin/bash
# round 6, the collection lease: GPU suite + fuzz on the final library, the profiles of tools/gpu_profile.sh, the perf floor
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python __graft_entry__.py smoke > $O/r06_smoke.txt 2>&1; tail -5 $O/r06_smoke.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r06_pytest_final.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_final.txt; tail -5 $O/r06_pytest_final.txt
( timeout 300 python tools/fuzz_gpu.py --seconds 150 --seed 620 --batch; timeout 200 python tools/fuzz_gpu.py --seconds 90 --seed 621 --many; timeout 200 python tools/fuzz_gpu.py --seconds 90 --seed 622 --many-channels ) 2>&1 | grep "fuzz:\|FAIL" | tee $O/r06_fuzz_final.txt
bash tools/gpu_profile.sh 06 > $O/r06_profile_log.txt 2>&1; tail -45 $O/r06_profile_log.txt
timeout 1200 python tools/perf_floor.py --measure --merge > $O/r06_perf_floor.txt 2>&1; tail -5 $O/r06_perf_floor.txt
cp profiles/perf_floor.json $O/perf_floor.json; cp profiles/pmc_traffic.json $O/pmc_traffic.json
