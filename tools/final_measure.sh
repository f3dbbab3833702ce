#!/bin/bash
# Round-end measurements on the GPU box (through gpurun, from the repo root): the profile recipes of profiles/README.md, the
# un-profiled bench lines and a batch sweep, into gpurun_out/ - copy the summaries to profiles/ afterwards
# (tools/collect_profiles.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RND=${RND:-r04}
bash profiles/prof_recipe.sh ${RND}_b256 --no-b1024 > gpurun_out/prof_b256.log 2>&1
bash profiles/prof_recipe.sh ${RND}_b1024 --batch 1024 > gpurun_out/prof_b1024.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg4_rf9 --workload cfg4_rf9 > gpurun_out/prof_cfg4.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg4_rf243 --workload cfg4_rf243 > gpurun_out/prof_cfg4b.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg5 --workload cfg5 > gpurun_out/prof_cfg5.log 2>&1
R3D_BF16X3=1 bash profiles/prof_recipe.sh ${RND}_b256_bf16x3 --no-b1024 > gpurun_out/prof_b3.log 2>&1
# the clip-sharded evaluation (BASELINE configs[2] shape): kernel trace only
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/${RND}_eval
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_eval/trace -- python $R/bench.py --mode eval > $R/gpurun_out/${RND}_eval/trace.log 2>&1
cp $(find $R/gpurun_out/${RND}_eval/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_eval/kernel_stats.csv
grep '^{' $R/gpurun_out/${RND}_eval/trace.log > $R/gpurun_out/${RND}_eval/bench_line.json
cd $R
python bench.py > gpurun_out/${RND}_bench_default.json 2> gpurun_out/${RND}_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${RND}_bench_driver_cmd.json 2> gpurun_out/${RND}_bench_driver_cmd.err
python bench.py --mode eval > gpurun_out/${RND}_bench_eval.json 2> gpurun_out/${RND}_bench_eval.err
for B in 1 8 32 64 128 512 2048 4096; do python bench.py --batch $B --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, l['ms_per_step'])"; done > gpurun_out/${RND}_batch_sweep.txt
