cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
R3D_BF16X3=$m rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/ic$m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-b1024 > $R/gpurun_out/ic$m.log 2>&1
R3D_BF16X3=$m rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/ia$m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-b1024 > $R/gpurun_out/ia$m.log 2>&1
done
ls $R/gpurun_out/ic0/*/ | head
