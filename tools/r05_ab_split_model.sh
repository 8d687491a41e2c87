# tools/r05_ab_split_model.sh -- round 5: the phase-group shares of wide-window launches by the generation model
# (default) against the doubling rule of rounds 3-4 (SPEEXHIP_SPLIT_MODEL=1); gpurun.
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { # ch pair q streams frames
for M in 1 new; do
if [ $M = new ]; then unset SPEEXHIP_SPLIT_MODEL; else export SPEEXHIP_SPLIT_MODEL=$M; fi
python bench.py --custom $1,$2,$3 --streams $4 --frames $5 --steps 6 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $1 $2 q$3 streams $4 frames $5 model=$M: %.1f us' % d['roofline']['launch_us'])"
done; }
for P in 48000,11025 44100,8000 48000,22050 44100,16000 44100,32000 32000,11025 96000,11025 96000,44100 88200,48000 32000,44100; do
for CH in ${CHANNELS:-1 2 4 6 8}; do for SHAPE in 1,1048576 8,131072 32,131072; do
one $CH $P 7 ${SHAPE%,*} ${SHAPE#*,}
done; done; done
for P in 48000,11025 44100,8000 44100,16000; do for CH in 1 2 4; do for SHAPE in 1,1048576 8,131072 32,131072; do
one $CH $P 10 ${SHAPE%,*} ${SHAPE#*,}
done; done; done
