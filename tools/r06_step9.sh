cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/fuzz_gpu.py --only 611002591 --many-channels 2>&1 | tail -2
timeout 400 python tools/fuzz_gpu.py --seconds 240 --seed 612 --many-channels 2>&1 | tail -4
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
