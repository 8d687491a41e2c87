cd ${GRAFT_REPO_ROOT:-/root/repo}
for E in "SPEEXHIP_MANY_LANES=1" "SPEEXHIP_MANY_LANES=2" "SPEEXHIP_MANY_LANES=1 SPEEXHIP_PIN_IN_COPY=0" "SPEEXHIP_MANY_LANES=2 SPEEXHIP_PIN_IN_COPY=0" "SPEEXHIP_MANY_LANES=2 SPEEXHIP_PIN_IN_COPY=0"; do echo "$E"; env $E SPEEXHIP_LIB_PATH=node-speex-resampler_amd/ab/libspeexhip_diag.so SPEEXHIP_PY_NO_TORCH=1 python tools/pinned_path_bench.py cfg2 2>/dev/null | python3 -c "
import json,sys;d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('many32_1048576'): print(k,{a:(b['ms'] if isinstance(b,dict) and 'ms' in b else b) for a,b in v.items() if a in ('pageable','pinned_in','pinned_in_pinned_out')})"; done
