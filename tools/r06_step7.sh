#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "window_layout or golden or baseline_configs or edge_cases or many_rates or float_entry or padded or mid_stream" > $O/r06_pytest_staging.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_staging.txt; tail -4 $O/r06_pytest_staging.txt
rm -f $O/r06_staging_ab.txt
for REP in 1 2; do
for C in "--config cfg4 --streams 32" "--config cfg4 --streams 1" "--custom 8,48000,44100,5 --io float --streams 32" "--custom 4,48000,44100,5 --streams 32" "--custom 8,32000,44100,7 --streams 32" "--custom 12,48000,44100,5 --streams 32" "--custom 2,48000,44100,5 --streams 32"; do
  for LIB in r05 r06; do
    P=$R/node-speex-resampler_amd/libspeexhip.so; [ $LIB = r05 ] && P=$R/node-speex-resampler_amd/ab/libspeexhip_r05.so
    SPEEXHIP_LIB_PATH=$P python bench.py $C --steps 40 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$LIB', '$C', 'launch_us', d['roofline']['launch_us'], d['roofline']['launch_us_min'], 'valu', d['valu']['frac'], 'parity', d['parity'].get('max_abs_diff_lsb', d['parity'].get('max_abs_diff')))" | tee -a $O/r06_staging_ab.txt
  done
done
done
