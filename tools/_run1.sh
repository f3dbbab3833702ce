cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r05_run1_tests.log 2>&1
tail -5 gpurun_out/r05_run1_tests.log
timeout 900 bash profiles/prof_recipe_eval.sh r05_eval_b4096 > gpurun_out/prof_eval.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r05_eval_b4096/pmc_table.txt
timeout 900 bash tools/b3_split_bound.sh run > gpurun_out/b3_split_bound.txt 2>gpurun_out/b3_split_bound.err
cat gpurun_out/b3_split_bound.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_run1_bench.json 2> gpurun_out/r05_run1_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_run1_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
