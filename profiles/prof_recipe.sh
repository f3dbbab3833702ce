set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof1
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc4.log 2>&1
find $O -name "*.csv" | head -30
du -sh $O
