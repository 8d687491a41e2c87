#!/bin/bash
# tools/ab.sh -- ONE parameterised same-box A/B runner (replaces the r04_* / r05_* one-off shells; their outputs stay
# in profiles/).  Runs a command once per setting of diagnostics switches on the DIAGNOSTICS build of the library
# (csrc/diag.h: ab/libspeexhip_diag.so through SPEEXHIP_LIB_PATH) inside one gpurun lease and prints a row per setting.
#
#   tools/ab.sh [-o out.txt] [-f jq-like python expr over the JSON line `d`] -- "ENV1=a ENV2=b" "ENV1=c" ... -- command...
#
# Each quoted group is one setting ("" = the defaults).  Example:
#   tools/ab.sh -f "d['roofline']['launch_us']" -- "" "SPEEXHIP_KSPLIT=0" "SPEEXHIP_R=10" -- python bench.py --no-cpu-baseline --reps 3
# With no -f the command's last line is printed as is.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=""; FILTER=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do
  case "$1" in
    -o) OUT=$2; shift 2;;
    -f) FILTER=$2; shift 2;;
    *) echo "tools/ab.sh: unknown option $1" >&2; exit 2;;
  esac
done
shift
SETTINGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETTINGS+=("$1"); shift; done
shift
[ $# -gt 0 ] || { echo "tools/ab.sh: no command" >&2; exit 2; }
cd $R
DIAG=$R/node-speex-resampler_amd/ab/libspeexhip_diag.so
[ -f $DIAG ] || { echo "tools/ab.sh: $DIAG not built (make -C node-speex-resampler_amd diag)" >&2; exit 2; }
emit() { if [ -n "$OUT" ]; then tee -a "$OUT"; else cat; fi; }
echo "# $(date -u +%FT%TZ) command: $*" | emit
for S in "${SETTINGS[@]}"; do
  LINE=$(env SPEEXHIP_LIB_PATH=$DIAG $S "$@" 2>/dev/null | tail -1)
  if [ -n "$FILTER" ]; then
    LINE=$(printf '%s' "$LINE" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print($FILTER)")
  fi
  printf '%-60s %s\n' "[${S:-defaults}]" "$LINE" | emit
done
