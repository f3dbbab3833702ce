cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r05_run6_tests.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r05_run6_tests.log | tail -5
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_run6_bench.json 2> gpurun_out/r05_run6_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_run6_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_b1024']['frac'], d['cfg4']['rf9']['value'], d['cfg5']['value'], d['bf16x3']['value'])"
