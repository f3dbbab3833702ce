#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lanes or chained or f32_error_budget or half_chip or two_streams" > gpurun_out/r06_lanes_tests.log 2>&1
grep -v "^parity" gpurun_out/r06_lanes_tests.log | tail -30
python bench.py --lanes 2 --half-chip-streams --no-shipped-cfgs --no-c1024 --no-bf16x3 --no-cpu-baseline --no-b1024 > gpurun_out/r06_bench_lanes2.json 2> gpurun_out/r06_bench_lanes2.err; tail -c 400 gpurun_out/r06_bench_lanes2.err
python bench.py --lanes 4 --no-shipped-cfgs --no-c1024 --no-bf16x3 --no-cpu-baseline --no-b1024 > gpurun_out/r06_bench_lanes4.json 2> gpurun_out/r06_bench_lanes4.err
python bench.py --mode eval --lanes 2 > gpurun_out/r06_bench_eval_lanes2.json 2> gpurun_out/r06_bench_eval_lanes2.err; tail -c 400 gpurun_out/r06_bench_eval_lanes2.err
python - <<'PY'
import json
for f in ("r06_bench_lanes2", "r06_bench_lanes4", "r06_bench_eval_lanes2"):
    try:
        l = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], json.dumps({k: l.get(k) for k in ("lanes_variant", "half_chip_streams_variant", "lanes") if l.get(k)})[:1500])
    except Exception as e:
        print(f, "unreadable", e)
PY
