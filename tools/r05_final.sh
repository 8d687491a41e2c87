# tools/r05_final.sh -- end of round 5 on one box: the GPU suite + fuzz (r05_validate.sh), the perf floor of this lease
# folded into a copy of profiles/perf_floor.json, the profiles of tools/gpu_profile.sh, the bench line in the driver's form.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
FUZZ_S=120 bash tools/r05_validate.sh 2>&1 | tail -14
python tools/perf_floor.py --measure --merge --forget-faster 0.85 --out gpurun_out/perf_floor.json > gpurun_out/r05_perf_floor_run.txt 2>&1; tail -60 gpurun_out/r05_perf_floor_run.txt
bash tools/gpu_profile.sh 05 > gpurun_out/r05_gpu_profile.log 2>&1; tail -5 gpurun_out/r05_gpu_profile.log
python bench.py > gpurun_out/r05_bench_driver_form.json 2> gpurun_out/r05_bench_driver_form.err; cat gpurun_out/r05_bench_driver_form.json
