#!/usr/bin/env python3
"""Coordinate search over the packer's cost-model constants (R3D_COST, r3d_schedule.cpp) on the GPU box.
Every trial is a fresh `bench.py --batch B` process per batch size; objective = sum over batch sizes of ms / baseline ms.
usage: python tools/tune_cost.py [rounds] > gpurun_out/tune_cost.log"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["iter", "fixed", "ks_iter", "ks_fixed", "first_extra", "first_extra_wide", "pair_scale", "nb_iter_extra", "nb_fixed_extra"]
BASE = [2200.0, 2500.0, 2600.0, 6800.0, 5.0, 7.0, 1.08, 60.0, 1500.0]      # = CostModel's defaults (r3d_schedule.cpp)
STEP = [0.06, 0.4, 0.1, 0.3, 0.6, 0.4, 0.06, 2.0, 0.6]          # relative steps
THRESH = 0.004      # sum over the batch sizes of ms / baseline ms must drop by this much (run-to-run noise is 0.1 - 0.2 % per size)
BATCHES = [256, 1024]


def run(cost, batch):
    env = dict(os.environ, R3D_COST=",".join("%g" % v for v in cost), R3D_USE_HOOKS_LIB="1")      # (R3D_COST is a hook_env switch: hooks build)
    best = 1e9
    for _ in range(3):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(batch), "--no-cpu-baseline", "--no-bf16x3",
                              "--no-shipped-cfgs", "--no-b1024", "--no-c1024", "--steps", "200", "--warmup", "10"], env=env, capture_output=True, text=True)
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        best = min(best, line["ms_per_step"])
    return best


def objective(cost, ref):
    ms = [run(cost, b) for b in BATCHES]
    return sum(m / r for m, r in zip(ms, ref)), ms


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ref = [run(BASE, b) for b in BATCHES]
print("baseline", BASE, ref, flush=True)
cur, cur_obj = list(BASE), float(len(BATCHES))
for rnd in range(rounds):
    for i, name in enumerate(NAMES):
        for sign in (+1, -1):
            trial = list(cur)
            trial[i] = max(cur[i] * (1 + sign * STEP[i]), 0.0)
            obj, ms = objective(trial, ref)
            print("round %d %s %+d -> %s obj %.4f ms %s%s" % (rnd, name, sign, ["%g" % v for v in trial], obj, ms, "  *" if obj < cur_obj - THRESH else ""), flush=True)
            if obj < cur_obj - THRESH:
                cur, cur_obj = trial, obj
                break
print("best", cur, cur_obj, flush=True)
