cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "window_layout_variants or int16_window_plan_serves or int16_window_on_small or wide_window_batches" 2>&1 | tail -3
for CFG in 16,96000,11025,7 16,96000,11025,10 12,96000,11025,10; do for IO in int16 float; do for SHAPE in 32,131072 1,1048576; do
python bench.py --custom $CFG --io $IO --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 4 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d.get('parity',{}); print('$CFG $IO $SHAPE: %.1f us valu %.3f path %d parity %s' % (d['roofline']['launch_us'], d['valu']['frac'], d['config']['fast_path'], p.get('max_abs_diff_lsb', p.get('max_abs_diff'))))"
done; done; done
timeout 200 python tools/fuzz_gpu.py --many-channels --seconds 100 --seed 555 2>&1 | tail -1
