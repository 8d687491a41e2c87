# tools/r05_ab_pp_deep.sh -- round 5: phase pairs on 40-tap banks (this build, resample_period_wide) against the build
# before (20-tap banks under the 80-SGPR cap: node-speex-resampler_amd/ab/libspeexhip_prev.so), same box; gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -x -q -k "phase_pair or every_golden or fast_fixed or window or layout" 2>&1 | tail -3
LP=node-speex-resampler_amd/ab/libspeexhip_prev.so; LN=node-speex-resampler_amd/libspeexhip.so
for P in 48000,11025 44100,8000 48000,22050 44100,16000 44100,32000 32000,11025 96000,11025 32000,44100 96000,44100; do
for CH in 1 2 3; do for SHAPE in 8,131072 32,131072 32,1048576; do
for L in $LP $LN; do
SPEEXHIP_LIB_PATH=$L python bench.py --custom $CH,$P,7 --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 8 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P streams ${SHAPE%,*} frames ${SHAPE#*,} lib=$(basename $L .so): %.1f us parity %s' % (d['roofline']['launch_us'], d.get('parity', {}).get('max_abs_diff_lsb')))"
done; done; done; done
