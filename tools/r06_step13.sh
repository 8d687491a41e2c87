#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "small_ratio or n_to_one or slide or golden or edge_cases or many_rates or fp64_accumulate_slide or baseline_configs or float_entry" > $O/r06_pytest_slide.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_slide.txt; tail -4 $O/r06_pytest_slide.txt
rm -f $O/r06_slide_staging_ab.txt
for REP in 1 2; do
for C in "--config f3" "--custom 2,24000,48000,5" "--custom 2,48000,24000,5" "--custom 1,16000,48000,7" "--custom 2,48000,16000,7" "--config cfg3" "--custom 1,48000,8000,7" "--custom 2,8000,48000,5" "--config f3 --io float" "--custom 3,24000,48000,5" "--config cfg4"; do
  for LIB in r05 r06; do
    P=$R/node-speex-resampler_amd/libspeexhip.so; [ $LIB = r05 ] && P=$R/node-speex-resampler_amd/ab/libspeexhip_r05.so
    SPEEXHIP_LIB_PATH=$P python bench.py $C --streams 32 --steps 40 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$LIB', '$C', 'launch_us', d['roofline']['launch_us'], d['roofline']['launch_us_min'], 'valu', d['valu']['frac'], 'parity', d['parity'].get('max_abs_diff_lsb', d['parity'].get('max_abs_diff')))" | tee -a $O/r06_slide_staging_ab.txt
  done
done
done
