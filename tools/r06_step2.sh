#!/bin/bash
# round 6, second lease: whole GPU suite on the default-mode / workers / N-API changes, perf floor rows of the default mode,
# host-many bench under 8 logical devices, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python __graft_entry__.py smoke > $O/r06_smoke.txt 2>&1; tail -4 $O/r06_smoke.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r06_pytest_gpu2.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_gpu2.txt
tail -30 $O/r06_pytest_gpu2.txt
timeout 900 python tools/perf_floor.py --measure --merge --only default > $O/r06_perf_floor_default.txt 2>&1; tail -12 $O/r06_perf_floor_default.txt
cp profiles/perf_floor.json $O/perf_floor.json
rm -f $O/r06_host_many_alias8.txt
for REP in 1 2; do for F in 16384 1048576; do
  for LIB in r05 r06; do
    P=$R/node-speex-resampler_amd/libspeexhip.so; [ $LIB = r05 ] && P=$R/node-speex-resampler_amd/ab/libspeexhip_r05.so
    [ -f $P ] || continue
    echo -n "$LIB frames=$F " | tee -a $O/r06_host_many_alias8.txt
    SPEEXHIP_LIB_PATH=$P SPEEXHIP_ALIAS_DEVICES=8 SPEEXHIP_DEVICES=all timeout 600 python tools/host_many_bench.py --streams 256 --frames $F --steps 16 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'min', d['ms_min'], 'devices', d['devices'])" | tee -a $O/r06_host_many_alias8.txt
  done
done; done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_form2.json 2> $O/r06_bench2.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/r06_bench_driver_form2.json'))
print(d['value'], d['ms_per_step'], d['roofline'], d['config']['mode']); print(d['pcie_peak']); print(d['end_to_end']); print(d['end_to_end_streams'])"
