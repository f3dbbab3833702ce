#!/bin/bash
# development aid: memory-path counters per launch (L1->L2 read latency, address translation, L2->fabric latency)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-b1024"
for m in 0 1; do
R3D_BF16X3=$m timeout 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/ma$m -- $B > $R/gpurun_out/ma$m.log 2>&1
R3D_BF16X3=$m timeout 150 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TA_BUSY_sum --output-format csv -d $R/gpurun_out/mc$m -- $B > $R/gpurun_out/mc$m.log 2>&1
R3D_BF16X3=$m timeout 150 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/mb$m -- $B > $R/gpurun_out/mb$m.log 2>&1
done
grep -l "exceeds the capabilities" $R/gpurun_out/m[abc][01].log
