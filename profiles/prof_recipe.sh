# usage (through gpurun, from the repo root): bash profiles/prof_recipe.sh <out-dir-name> [bench.py arguments]
# e.g. bash profiles/prof_recipe.sh r02_b256 --no-b1024      bash profiles/prof_recipe.sh r02_b1024 --batch 1024
# One kernel trace with --stats, then four separate --pmc passes (never combined with another trace domain).
set -x
NAME=${1:-prof}; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-c1024 $ARGS > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-c1024 $ARGS > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-c1024 $ARGS > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-c1024 $ARGS > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-c1024 $ARGS > $O/pmc4.log 2>&1
find $O -name "*.csv" | head -30
# the summaries that get committed: per-kernel stats of the traced run, the bench line of that run, the PMC table
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep '^{' $O/trace.log > $O/bench_line.json
python $R/profiles/summarize_pmc.py $O > $O/pmc_table.txt   # (+ $O/pmc_summary.json)
du -sh $O
