# usage (through gpurun, from the repo root): bash profiles/prof_recipe_eval.sh <out-dir-name>        e.g. r05_eval_b4096
# The clip-sharded evaluation (bench.py --mode eval, BASELINE configs[2]'s shape) under rocprofv3: one kernel trace with --stats of
# the whole 240-clip pass, then the four separate --pmc passes (never combined with another trace domain) on a SHORT run
# (8 clips) whose last r3d_forward_clip_f32 dispatches are the 4096-window clip call bench.py's eval roofline times - the
# dispatch profiles/summarize_pmc.py takes (the last one of that name), so that the counters are per launch of exactly
# the call `roofline.achieved` is quoted on.
set -x
NAME=${1:-r05_eval_b4096}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --mode eval > $O/trace.log 2>&1
PM="python $R/bench.py --mode eval --clips 8 --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc1 -- $PM > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc2 -- $PM > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $PM > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc4 -- $PM > $O/pmc4.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep '^{' $O/trace.log > $O/bench_line.json
python $R/profiles/summarize_pmc.py $O > $O/pmc_table.txt   # (+ $O/pmc_summary.json)
du -sh $O
