#!/bin/bash
# coordinate search over the packer's cost constants on the final kernels (tools/tune_cost.py, hooks build), bounded in time
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout ${2:-1800} python tools/tune_cost.py ${1:-2} > gpurun_out/r06_tune_cost.log 2> gpurun_out/r06_tune_cost.err
tail -40 gpurun_out/r06_tune_cost.log
