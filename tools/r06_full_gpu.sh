#!/bin/bash
# full GPU pass of the round: every gpu test, the default bench line, the driver's command, the eval mode, the chain A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r06_gputests.log 2>&1
grep -v "^parity test_\|^parity RELAXED\|^parity f32" gpurun_out/r06_gputests.log | tail -25
grep "^parity f32-budget" gpurun_out/r06_gputests.log > gpurun_out/r06_f32_budget.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err; tail -c 600 gpurun_out/r06_bench_driver_cmd.err
python bench.py --mode eval > gpurun_out/r06_bench_eval.json 2> gpurun_out/r06_bench_eval.err; tail -c 300 gpurun_out/r06_bench_eval.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_driver_cmd.json", "gpurun_out/r06_bench_eval.json"):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l.get("roofline", {}).get("frac"), {k: (v.get("value") if isinstance(v, dict) else v) for k, v in l.items() if k in ("c1024", "cfg5", "bf16x3", "cpu_baseline", "eval_pass", "frac_of_fp32_mfma_peak")})
    except Exception as e:
        print(f, "unreadable", e)
PY
