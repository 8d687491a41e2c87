# tools/r05_i16_rides.sh -- round 5: layouts whose float-window plan fits under a ninth of a tile but whose int16-window plan
# does not: before (exact kernel; SPEEXHIP_MIN_FILL=9 cannot reproduce it -- the library of the commit before) and now; gpurun
cd ${GRAFT_REPO_ROOT:-/root/repo}
for CFG in 8,96000,11025,10 8,96000,11025,8 7,96000,11025,10 1,64000,11025,7 2,64000,11025,7 8,64000,11025,5; do for SHAPE in 32,131072 1,1048576; do for IO in int16 float; do
python bench.py --custom $CFG --io $IO --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 6 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$CFG $SHAPE $IO: %.1f us path %d valu %.3f parity %s' % (d['roofline']['launch_us'], d['config']['fast_path'], d['valu']['frac'], d.get('parity', {})))"
done; done; done
