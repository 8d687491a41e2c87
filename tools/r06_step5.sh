#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r06_pytest_gpu5.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_gpu5.txt; tail -6 $O/r06_pytest_gpu5.txt
timeout 400 python tools/fuzz_gpu.py --seconds 200 --seed 601 > $O/r06_fuzz_a.txt 2>&1; tail -3 $O/r06_fuzz_a.txt
timeout 300 python tools/fuzz_gpu.py --seconds 100 --seed 602 --many > $O/r06_fuzz_b.txt 2>&1; tail -3 $O/r06_fuzz_b.txt
rm -f $O/r06_fold_ab.txt
for C in 1,88000,8000,5 2,56000,48000,4 2,72000,16000,7 1,64000,12000,7 2,200000,8000,5 1,56000,48000,10; do
  for LIB in r05 r06; do
    P=$R/node-speex-resampler_amd/libspeexhip.so; M=fast_fixed; [ $LIB = r05 ] && P=$R/node-speex-resampler_amd/ab/libspeexhip_r05.so
    [ -f $P ] || continue
    for S in 1 32; do
    SPEEXHIP_LIB_PATH=$P python bench.py --custom $C --streams $S --frames 131072 --steps 30 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$LIB', '$C', 'streams', $S, 'fast_path', d['config']['fast_path'], 'launch_us', d['roofline']['launch_us'], 'valu', d['valu']['frac'], 'parity', d['parity'].get('max_abs_diff_lsb'), d['parity'].get('mismatch_rate'))" | tee -a $O/r06_fold_ab.txt
    done
  done
done
