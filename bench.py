#!/usr/bin/env python3
"""Throughput of the Ray3D lifting forward pass on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 17-joint, 243-frame ray-encoded windows, batch 256
per GPU, forward only, pos + trj networks (C=256, latent 256, stage 3, camera embedding on),
random-init (deterministic synthetic) weights, fp32.  A step = one pass of the whole path over one
batch already resident in HBM.  Weak scaling: every rank lifts its own 256-window batch, no
collective on the data path (the only collectives are the timing barrier / max).

Prints ONE JSON line (rank 0).  Besides the driver's keys it carries
  roofline      - the dominant kernel's measured rate (HIP events on the launch stream, inside
                  this process) against the gfx950 fp32-MFMA peak, plus the HBM view;
  cpu_baseline  - the PyTorch-CPU port of the same module graph (oracle/torch_port.py) timed on
                  this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch   # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0
BATCH = 256
ARCH = "3,3,3,3,3"


def build(device, arch=ARCH):
    import ray3d_amd
    from ray3d_amd import synth
    mc = ray3d_amd.default_model_config(ARCHITECTURE=arch)
    fac = ray3d_amd.Model(mc, {}, is_train=False)
    pos, trj = fac.get_pos_model(), fac.get_trj_model()
    states = {}
    for m, kind, seed in ((pos, "pos", 1), (trj, "trj", 2)):
        cfg = ray3d_amd.config_from_dicts(mc, kind)
        st = synth.synth_state(cfg, seed=seed)
        states[kind] = (cfg, st)
        ray3d_amd.load_weight(m, {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        m.eval()
    lifter = ray3d_amd.Ray3DLifter(pos, trj).eval()
    return lifter, states


def cpu_baseline(states, x, p, budget_s=20.0):
    """oracle/torch_port.py (the ATen CPU kernels the reference would run) on this host's cores.
    Thread counts above the cgroup's share thrash badly, so a short ladder of thread counts is tried
    inside the time budget and the fastest one is reported (cores = threads actually used)."""
    from oracle import torch_port
    sds = {k: {n: torch.from_numpy(np.asarray(v)) for n, v in st.items()} for k, (_, st) in states.items()}
    xt, pt = torch.from_numpy(x), torch.from_numpy(p)

    def run():
        with torch.no_grad():
            return torch_port.forward(states["pos"][0], sds["pos"], xt, pt) + \
                torch_port.forward(states["trj"][0], sds["trj"], xt, pt)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ladder = sorted(set(t for t in (8, 16, 32, 64, 128) if t <= avail) | ({avail} if avail < 8 else set()))
    best, best_threads, runs, t_all = float("inf"), ladder[0], 0, time.perf_counter()
    for threads in ladder:
        if time.perf_counter() - t_all > budget_s:
            break
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        run()                                   # warm-up at this thread count
        if time.perf_counter() - t0 > budget_s / 4:
            continue                            # hopeless at this count; do not burn the budget
        for _ in range(3):
            t0 = time.perf_counter()
            run()
            dt = time.perf_counter() - t0
            runs += 1
            if dt < best:
                best, best_threads = dt, threads
    return {"value": round(x.shape[0] / best, 1), "unit": "poses/s", "cores": best_threads, "kind": "port",
            "host_cpus": avail,
            "sample": "best of %d timed forwards (thread ladder %s) of one %d-window batch (pos+trj, RF 243) "
                      "through oracle/torch_port.py (PyTorch-CPU functional port of the reference graph)"
                      % (runs, ladder, x.shape[0])}


def roofline(lifter, x, p, reps=5):
    """Per-launch HIP events (bracketing each launch on its stream) -> dominant kernel's rate."""
    agg = {}
    lifter.profile(x, p)
    pair_ms = []
    for _ in range(reps):
        for r in lifter.profile(x, p):
            if r["kernel"] == "r3d_event_pair":        # the empty bracket: what the two event records cost
                pair_ms.append(r["ms"])
                continue
            a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
    pair = sorted(pair_ms)[len(pair_ms) // 2] if pair_ms else 0.0
    if os.environ.get("R3D_DUMP_LAUNCHES"):
        for r in lifter.profile(x, p):
            print("launch %2d %-20s blocks %5d  %8.1f us  %7.2f GFLOP  %6.1f TFLOP/s" % (
                r["stage"], r["kernel"], r["blocks"], r["ms"] * 1e3, r["flops"] / 1e9,
                r["flops"] / max(r["ms"], 1e-9) / 1e9), file=sys.stderr)
    name = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[name]
    raw_ms = d["ms"]
    # a bracket = the kernel + the event records around it; the empty bracket measures the latter (median of the
    # reps) and is taken off every launch - rocprofv3's kernel-trace durations (profiles/) have no such term
    d = dict(d, ms=max(d["ms"] - pair * d["launches"], 0.5 * d["ms"]))
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if measured
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(name) if tj.get("batch") == x.shape[0] else None   # measured at that batch size only
        except Exception:
            traffic = None
    out = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
           "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
           "launches_per_step": d["launches"] // reps,
           "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
           "avg_launch_us_with_events": round(raw_ms / d["launches"] * 1e3, 2), "event_pair_us": round(pair * 1e3, 2),
           "flops_per_launch": d["flops"] / d["launches"],
           "hbm_view": {"algorithmic_GBps": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                        "frac": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
           "all_kernels_us_per_step": {k: round(v["ms"] / reps * 1e3, 1) for k, v in agg.items()}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-stream", action="store_true", help="also time Ray3DLifter.forward_overlapped (two half batches on two streams)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"

    from ray3d_amd import synth
    lifter, states = build(dev)
    cfg = states["pos"][0]
    x_np = synth.synth_rays(args.batch, cfg, seed=100 + rank)
    p_np = synth.synth_param(args.batch, seed=0, vary=False)
    x, p = torch.from_numpy(x_np).to(dev), torch.from_numpy(p_np).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        # one-time set-up of the library for this batch size (launch plan, tile schedule upload, workspace
        # allocation) - not a warm-up step
        lifter(x, p)
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            out = lifter(x, p)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = lifter(x, p)
        barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        n_gpus = world
        value = n_gpus * args.batch * args.steps / elapsed
        line = {
            "metric": "lifted poses/sec (17-joint, 243-frame window)",
            "value": round(value, 1), "unit": "poses/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 17-joint 243-frame windows, batch %d per GPU, "
                                   "forward-only pos+trj (C=256, latent 256, stage 3, camera embedding)" % args.batch,
                       "batch_per_gpu": args.batch, "receptive_field": 243, "joints": 17,
                       "parallelism": "dp%d (independent window batches, no data-path collective)" % n_gpus},
        }
        with torch.no_grad():
            line["roofline"] = roofline(lifter, x, p)
        if n_gpus == 1 and args.two_stream:
            # informative, not the headline (opt-in so that the default command - the one profiles/prof_recipe.sh
            # traces - launches full batches only): the same batch as two independent half batches on two HIP streams
            # (Ray3DLifter.forward_overlapped) - launch tails and small levels of one half overlap the other's work
            with torch.no_grad():
                for _ in range(args.warmup):
                    lifter.forward_overlapped(x, p)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    lifter.forward_overlapped(x, p)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            line["two_stream_variant"] = {"value": round(args.batch * args.steps / dt, 1), "unit": "poses/s",
                                          "ms_per_step": round(dt / args.steps * 1e3, 4),
                                          "note": "same work as `value`, issued as 2 half batches on 2 streams"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(states, x_np, p_np)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
