#!/bin/bash
# Round-end measurements on the GPU box (through gpurun, from the repo root): the profile recipes of profiles/README.md, the
# three un-profiled bench lines and a batch sweep, into gpurun_out/ - copy the summaries to profiles/ afterwards.
cd /root/repo
bash profiles/prof_recipe.sh r03_b256 --no-b1024 > gpurun_out/prof_b256.log 2>&1
bash profiles/prof_recipe.sh r03_b1024 --batch 1024 > gpurun_out/prof_b1024.log 2>&1
bash profiles/prof_recipe.sh r03_cfg4_rf9 --workload cfg4_rf9 > gpurun_out/prof_cfg4.log 2>&1
bash profiles/prof_recipe.sh r03_cfg4_rf243 --workload cfg4_rf243 > gpurun_out/prof_cfg4b.log 2>&1
bash profiles/prof_recipe.sh r03_cfg5 --workload cfg5 > gpurun_out/prof_cfg5.log 2>&1
R3D_BF16X3=1 bash profiles/prof_recipe.sh r03_b256_bf16x3 --no-b1024 > gpurun_out/prof_b3.log 2>&1
bash profiles/prof_recipe.sh r03_b1 --batch 1 > gpurun_out/prof_b1.log 2>&1
cd /root/repo
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_cmd.json 2> gpurun_out/r03_bench_driver_cmd.err
python bench.py --mode eval > gpurun_out/r03_bench_eval.json 2> gpurun_out/r03_bench_eval.err
for B in 1 2 4 8 16 32 48 64 96 128 512; do python bench.py --batch $B --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, l['ms_per_step'])"; done > gpurun_out/r03_batch_sweep.txt
