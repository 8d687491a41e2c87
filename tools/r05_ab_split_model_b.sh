# tools/r05_ab_split_model_b.sh -- the batch rows of tools/r05_ab_split_model.sh once more (final form of the model); gpurun
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { for M in 1 new; do
if [ $M = new ]; then unset SPEEXHIP_SPLIT_MODEL; else export SPEEXHIP_SPLIT_MODEL=$M; fi
python bench.py --custom $1,$2,$3 --streams $4 --frames $5 --steps 8 --warmup 2 --reps 3 --preheat-ms 30 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $1 $2 q$3 streams $4 frames $5 model=$M: %.1f us' % d['roofline']['launch_us'])"
done; }
for P in 32000,11025 96000,11025 48000,11025 44100,8000 44100,16000; do for CH in 1 2 4 8; do for SHAPE in 8,131072 32,131072; do
one $CH $P 7 ${SHAPE%,*} ${SHAPE#*,}
done; done; done
for P in 48000,11025 44100,8000; do for CH in 2 4; do for SHAPE in 8,131072 32,131072; do one $CH $P 10 ${SHAPE%,*} ${SHAPE#*,}; done; done; done
