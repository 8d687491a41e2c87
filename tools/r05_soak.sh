# tools/r05_soak.sh -- a longer fuzz soak on the round's final library: host calls, batched device calls, many-states calls (gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05_soak
timeout 700 python tools/fuzz_gpu.py --seconds 400 --seed 9101 > gpurun_out/r05_soak/host.txt 2>&1; tail -n 1 gpurun_out/r05_soak/host.txt
timeout 700 python tools/fuzz_gpu.py --batch --seconds 400 --seed 9102 > gpurun_out/r05_soak/batch.txt 2>&1; tail -n 1 gpurun_out/r05_soak/batch.txt
timeout 500 python tools/fuzz_gpu.py --many --seconds 240 --seed 9103 > gpurun_out/r05_soak/many.txt 2>&1; tail -n 1 gpurun_out/r05_soak/many.txt
