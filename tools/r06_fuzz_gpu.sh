#!/bin/bash
# more seeds of the seeded-random-configuration test and every window at the plan-switch batch sizes, on the final kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_configs.py ${1:-1000} ${2:-80} > gpurun_out/r06_fuzz.log 2>&1; tail -3 gpurun_out/r06_fuzz.log
timeout 900 python tools/plan_edges.py > gpurun_out/r06_plan_edges.log 2>&1; tail -2 gpurun_out/r06_plan_edges.log
