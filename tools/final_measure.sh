#!/bin/bash
# Round-end measurements on the GPU box (through gpurun, from the repo root): the profile recipes of profiles/README.md, the
# un-profiled bench lines, a batch sweep and the per-tile timelines, into gpurun_out/ - copy the summaries to profiles/ afterwards
# (tools/collect_profiles.sh).   RND=r05 bash tools/final_measure.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RND=${RND:-r06}
bash profiles/prof_recipe.sh ${RND}_b256 --no-b1024 > gpurun_out/prof_b256.log 2>&1
bash profiles/prof_recipe.sh ${RND}_b1024 --batch 1024 > gpurun_out/prof_b1024.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg4_rf9 --workload cfg4_rf9 > gpurun_out/prof_cfg4.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg4_rf243 --workload cfg4_rf243 > gpurun_out/prof_cfg4b.log 2>&1
bash profiles/prof_recipe.sh ${RND}_cfg5 --workload cfg5 > gpurun_out/prof_cfg5.log 2>&1
R3D_BF16X3=1 bash profiles/prof_recipe.sh ${RND}_b256_bf16x3 --no-b1024 > gpurun_out/prof_b3.log 2>&1
# the clip-sharded evaluation (BASELINE configs[2] shape): kernel trace of the whole pass + PMC passes of the 4096-window clip call
bash profiles/prof_recipe_eval.sh ${RND}_eval_b4096 > gpurun_out/prof_eval.log 2>&1
# the 1024-channel model (level-by-level form): kernel trace only - its twelve r3d_gemm_f32 launches per step have no single "the" dispatch for a PMC row
(cd /tmp && export TMPDIR=/tmp && mkdir -p $R/gpurun_out/${RND}_c1024 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_c1024/trace -- python $R/bench.py --c1024 --steps 40 --warmup 5 > $R/gpurun_out/${RND}_c1024/trace.log 2>&1; cp $(find $R/gpurun_out/${RND}_c1024/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_c1024/kernel_stats.csv; grep '^{' $R/gpurun_out/${RND}_c1024/trace.log > $R/gpurun_out/${RND}_c1024/bench_line.json)
cd $R
python bench.py --half-chip-streams --lanes 2 > gpurun_out/${RND}_bench_default.json 2> gpurun_out/${RND}_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${RND}_bench_driver_cmd.json 2> gpurun_out/${RND}_bench_driver_cmd.err
python bench.py --mode eval --lanes 2 > gpurun_out/${RND}_bench_eval.json 2> gpurun_out/${RND}_bench_eval.err
# the driver's launcher with one rank: the line then carries the clip-sharded `eval_pass` object (RCCL, world size 1)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-shipped-cfgs --no-c1024 --no-bf16x3 --no-b1024 > gpurun_out/${RND}_bench_launcher_n1.json 2> gpurun_out/${RND}_bench_launcher_n1.err
for B in 1 8 32 64 128 192 512 2048 4096; do python bench.py --batch $B --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --no-c1024 --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, l['ms_per_step'])"; done > gpurun_out/${RND}_batch_sweep.txt
# per-tile timelines (timing build of the library: tools/build_probe.sh or the hipcc line in tools/README.md)
if [ -f tools/libray3d_hip_timing.so ]; then
  for B in 256 1024; do
    mkdir -p gpurun_out/${RND}_b$B
    R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=all R3D_TIMING_DUMP=gpurun_out/gantt_$B.txt python tools/stage_times.py $B 1 > gpurun_out/gantt_$B.log 2>&1
    python tools/fwd_gantt.py gpurun_out/gantt_$B.txt > gpurun_out/${RND}_b$B/fwd_gantt.txt 2>&1
  done
fi
