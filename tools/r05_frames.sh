# tools/r05_frames.sh -- frames of 10 / 12 / 16 channels on ISA loops (kernels_period_frames.hip): parity, timing, fuzz; gpurun
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "window_layout_variants or int16_window_on_small or wide_window_batches or tap_range_shares" 2>&1 | tail -3
for CH in 10 12 16; do for P in 44100,48000 48000,44100 48000,11025 44100,8000; do for IO in int16 float; do for SHAPE in 32,131072 1,1048576; do
python bench.py --custom $CH,$P,7 --io $IO --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 6 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d.get('parity',{}); print('ch $CH $P $IO $SHAPE: %.1f us valu %.3f parity %s' % (d['roofline']['launch_us'], d['valu']['frac'], p.get('max_abs_diff_lsb', p.get('max_abs_diff'))))"
done; done; done; done
timeout 300 python tools/fuzz_gpu.py --many-channels --seconds 150 --seed 4242 2>&1 | tail -2
