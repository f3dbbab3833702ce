#!/bin/bash
# the round's last GPU pass: the whole GPU suite, smoke, then the bench lines on the committed pmc.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r06_final_tests.log 2>&1
grep -v "^parity" gpurun_out/r06_final_tests.log | tail -15
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.log 2>&1; tail -2 gpurun_out/r06_final_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
python bench.py --half-chip-streams --lanes 2 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --lanes 4 --no-shipped-cfgs --no-c1024 --no-bf16x3 --no-cpu-baseline --no-b1024 > gpurun_out/r06_bench_lanes4.json 2> gpurun_out/r06_bench_lanes4.err
python bench.py --mode eval --lanes 2 > gpurun_out/r06_bench_eval.json 2> gpurun_out/r06_bench_eval.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_launcher_n1.json 2> gpurun_out/r06_bench_launcher_n1.err
python - <<'PY'
import json
for f in ("driver_cmd", "default", "lanes4", "eval", "launcher_n1"):
    try:
        l = json.loads(open("gpurun_out/r06_bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"].get("pmc_source", "")[:60], sorted(k for k in l if k not in ("metric", "unit", "config")))
    except Exception as e:
        print(f, "unreadable", e)
PY
