# tools/r05_w16_64.sh -- round 5: parity + timing of the fp64 period kernel over the int16 window (gpurun).
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "fp64 or window or every_golden or fast_fixed or phase_pair or layout" 2>&1 | tail -3
SPEEXHIP_W16_ALWAYS=1 python -m pytest tests -m gpu -x -q -k "fp64 or window or every_golden or fast_fixed or phase_pair or layout" 2>&1 | tail -3
timeout 400 python tools/fuzz_gpu.py --seconds 120 2>&1 | tail -3
timeout 400 python tools/fuzz_gpu.py --seconds 60 --many 2>&1 | tail -3
bash tools/r05_q10_decim.sh
