cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in 0 1; do SPEEXHIP_NAPI_COPY=$C node --expose-gc tools/node_pinned_ab.js 2>&1 | tail -2 | tee -a gpurun_out/r06_node_pinned_ab.txt; done
