# tools/r05_wide_q7.sh -- round 5: where the time of the num = 1280 decimators (32k / 96k -> 11.025k, q7) goes: phases
# of the kernel (SPEEXHIP_SKIP) and the planner's knobs (gpurun).
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env..., -- ch pair
  local label="$1"; shift
  env "$@" python bench.py --custom $CH,$P,7 --streams 32 --frames 131072 --steps 8 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline --no-parity $EXTRA 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P %-28s %.1f us' % ('$label', d['roofline']['launch_us']))"
}
for P in 32000,11025 96000,11025 48000,11025; do for CH in 1 2 4; do
run "rule" A=1
run "staging only" SPEEXHIP_SKIP=12
run "FIR only" SPEEXHIP_SKIP=10
run "stores only" SPEEXHIP_SKIP=6
run "PP=0" SPEEXHIP_PP=0
run "PP=1" SPEEXHIP_PP=1
run "W16=0" SPEEXHIP_W16_ALWAYS=0 SPEEXHIP_NO_W16=1
run "SPLITS=1" SPEEXHIP_SPLITS=1
run "SPLITS=2" SPEEXHIP_SPLITS=2
run "SPLITS=8" SPEEXHIP_SPLITS=8
run "KSPLIT=2" SPEEXHIP_KSPLIT=2
run "KSPLIT=4" SPEEXHIP_KSPLIT=4
run "KSPLIT=8" SPEEXHIP_KSPLIT=8
EXTRA="--mode exact" run "exact kernel" A=1
done; done
