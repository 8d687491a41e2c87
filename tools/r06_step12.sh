cd ${GRAFT_REPO_ROOT:-/root/repo}
LINES_ONLY=1 bash tools/gpu_profile.sh 06 2>&1 | tail -40
bash tools/r06_step11.sh
