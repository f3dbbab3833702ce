#!/usr/bin/env python3
"""Throughput of the Ray3D lifting forward pass on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode windows|eval] [--batch B]

--mode windows (default; BASELINE.json configs[1]): synthetic 17-joint, 243-frame ray-encoded windows, batch 256 per
GPU, forward only, pos + trj networks (C=256, latent 256, stage 3, camera embedding on), random-init (deterministic
synthetic) weights, fp32.  A step = one pass of the whole path over one batch already resident in HBM.  Weak scaling:
every rank lifts its own batch, no collective on the data path (only the timing barrier / max).

--mode eval (BASELINE.json configs[2]): the Human3.6M evaluation SHAPE (the data set is not in the image) - 240
synthetic clips of U(1000, 6000) frames, four cameras, fifteen actions - sharded over the ranks as whole clips
(longest first), lifted with in-kernel sliding windows, errors summed on the device, ONE RCCL all_gather of the
per-clip partial rows per pass.  A step = one pass over the whole clip set (strong scaling: the set is fixed);
the gathered MPJPE is part of the line and must not depend on N.

--workload cfg4_rf9 | cfg4_rf243 | cfg5 (windows mode): the configurations the reference SHIPS instead of configs[1] -
BASELINE configs[3] (cfg_ray3d_3dhp_stage3: J = 17, ARCHITECTURE '3,3' = RF 9, and the same at RF 243; pixel keypoints in,
every window with its own camera drawn from the 14 MPI-INF-3DHP cameras, rays encoded in the first-level gather; 1024
windows per step) and configs[4] (14 joints, RF 9, 4096 windows per step - 512 per GPU under --gpus 8 -, one camera of
the 342-camera augmentation grid of data/camera_augmentation.py:637-642 per window).  The default line carries them as
the secondary objects `cfg4` and `cfg5`.

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks through
torch.distributed.run (one process per GPU, RCCL) and fails loudly when fewer GPUs are visible.

Prints ONE JSON line (rank 0).  Besides the driver's keys it carries
  roofline        the dominant kernel's rate against the gfx950 fp32-MFMA peak.  Per-launch HIP events on the launch
                  stream give every launch's duration; they are scaled so that their sum per step equals the step time
                  of the timed region (HIP events around it, same stream) - bracketing every launch slows the chip's
                  clock and adds gaps, and a kernel cannot take longer than the step it is part of;
  roofline_b1024  the same at north_star's 1024-window batch (windows mode, N = 1);
  parity_max_abs_err   the HIP path against tests/golden/model_j17_rf243_s3.npz (outputs of the REFERENCE's PyTorch-CPU
                  forward for the very weights this bench builds), checked BEFORE anything is timed: > 1e-4 aborts;
  cfg4, cfg5      poses/s + roofline of the shipped RF-9 configurations (see --workload);
  cpu_baseline    the PyTorch-CPU port of the same module graph (oracle/torch_port.py) and the C restatement
                  (oracle/ray3d_oracle.c) timed on this host's cores on a bounded sample of the same workload (rank 0, every N);
  eval_pass       (windows mode, whenever a process group exists - the driver's --gpus N > 1 command, or any run under a
                  launcher) north_star's multi-GPU split: the --mode eval workload, clips sharded over the ranks, one RCCL
                  all_gather of the per-clip rows per pass: poses/s (strong scaling), shard_frames, pass_ms_per_rank,
                  pass_ms_imbalance, all_gather_ms, the gathered MPJPE + checksum (independent of N), world_size_observed,
                  cross_rank_rows_bit_equal (every rank re-lifts one clip of its neighbour's shard: same bits).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch   # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_BF16X3_TFLOPS = 2500.0 / 6.0   # dense bf16 MFMA peak / six bf16 products per fp32 product (r3d_config.bf16x3)
PEAK_HBM_GBS = 8000.0
PARITY_FIXTURE = os.path.join(ROOT, "tests", "golden", "model_j17_rf243_s3.npz")
PARITY_ATOL = 1e-4              # north_star: fp32 3D joint positions within 1e-4 abs of the reference's CPU forward
BATCH = 256
ARCH = "3,3,3,3,3"
EVAL_CLIPS = 240                # 2 subjects x 15 actions x 2 sub-actions x 4 cameras (cfg 3)


def build(device, arch=ARCH, bf16x3=False, **over):
    import ray3d_amd
    from ray3d_amd import synth
    mc = ray3d_amd.default_model_config(ARCHITECTURE=arch, BF16X3=bf16x3, **over)
    fac = ray3d_amd.Model(mc, {}, is_train=False)
    pos, trj = fac.get_pos_model(), fac.get_trj_model()
    states = {}
    for m, kind, seed in ((pos, "pos", 1), (trj, "trj", 2)):
        cfg = ray3d_amd.config_from_dicts(mc, kind)
        st = synth.synth_state(cfg, seed=seed)
        states[kind] = (cfg, st)
        ray3d_amd.load_weight(m, {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        m.eval()
    lifter = ray3d_amd.Ray3DLifter(pos.to(device), trj.to(device)).eval()
    return lifter, states


def cpu_baseline(states, x, p, budget_s=20.0):
    """oracle/torch_port.py (the ATen CPU kernels the reference would run) on this host's cores, and the C
    restatement beside it.  Thread counts above the cgroup's share thrash badly, so a short ladder of thread counts is
    tried inside the time budget and the fastest one is reported (cores = threads actually used)."""
    from oracle import oracle, torch_port
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
    out = {"value": round(x.shape[0] / best, 1), "unit": "poses/s", "cores": best_threads, "kind": "port",
           "host_cpus": avail,
           "sample": "best of %d timed forwards (thread ladder %s) of one %d-window batch (pos+trj, RF 243) "
                     "through oracle/torch_port.py (PyTorch-CPU functional port of the reference graph)"
                     % (runs, ladder, x.shape[0])}
    # the C restatement (OpenMP, double accumulation, un-folded BatchNorm: a checker, not a tuned GEMM) on a smaller sample
    ns = min(64, x.shape[0])
    threads = min(avail, 64)
    t0 = time.perf_counter()
    oracle.forward(states["pos"][0], states["pos"][1], x[:ns], p[:ns], threads=threads)
    oracle.forward(states["trj"][0], states["trj"][1], x[:ns], p[:ns], threads=threads)
    dt = time.perf_counter() - t0
    out["c_oracle"] = {"value": round(ns / dt, 1), "unit": "poses/s", "cores": threads,
                       "sample": "one forward of %d windows (pos+trj, RF 243) through oracle/ray3d_oracle.c" % ns}
    return out


def settle_clocks(fn, dev, group=10, max_groups=60, tol=0.01):
    """Runs the step until its time stops falling: a process starts with the GPU in a low power state and the clocks need
    ~30 ms of load to reach their operating point (tools/clock_ramp.py: 0.75 ms/step for steps 0-5, 0.67 for 5-25, 0.60
    from step 50 on at 256 windows) - longer than the W warm-up steps a caller may ask for.  Groups of `group` steps
    until two consecutive groups agree within `tol` (at most group * max_groups steps).  Untimed, before the W warm-up
    steps; the number of steps is reported (config.clock_settle_steps)."""
    prev, n = None, 0
    for _ in range(max_groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(group):
            fn()
        e1.record(torch.cuda.current_stream(dev))
        e1.synchronize()
        n += group
        t = e0.elapsed_time(e1)
        if prev is not None and abs(t - prev) <= tol * prev:
            break
        prev = t
    return n


def timed_steps(fn, steps, warmup, barrier, dev):
    """W untimed + exactly K timed steps between barriers; host wall time and the device time between two HIP events
    recorded on the launch stream around the same K steps."""
    on_gpu = torch.device(dev).type == "cuda"          # (the clip-sharded pass's host logic is also run on CPU ranks over gloo: tests/test_host.py)
    e0, e1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if on_gpu else (None, None)
    out = None
    for _ in range(warmup):
        out = fn()
    barrier()
    t0 = time.perf_counter()
    if on_gpu:
        e0.record(torch.cuda.current_stream(dev))
    for _ in range(steps):
        out = fn()
    if on_gpu:
        e1.record(torch.cuda.current_stream(dev))
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, (e0.elapsed_time(e1) * 1e-3 if on_gpu else elapsed), out


def pmc_figures(key, kernel):
    """What the committed rocprofv3 PMC passes say about `kernel` on workload `key` (profiles/pmc.json, written from
    profiles/<round>_<key>/ by profiles/make_pmc_json.py): HBM bytes per launch, MFMA-busy fraction, executed FLOPs."""
    f = os.path.join(ROOT, "profiles", "pmc.json")
    try:
        return json.load(open(f))["workloads"][key]["kernels"][kernel]
    except Exception:                                  # noqa: BLE001 - no profile of this workload: the line says null
        return None


def roofline(lifter, x, p, step_ms, reps=5, fn=None, batch=None, key=None):
    """Per-launch HIP events (bracketing each launch on its stream) -> the dominant kernel's rate.
    `step_ms`: device time of one step of the timed region; the launch durations are scaled to sum to it.
    `fn`: the forward to profile when it is not lifter(x, p) (UV mode); `key`: the workload's name in profiles/pmc.json.
    The peak is the one of the arithmetic the handles actually run (r3d_precision): an environment override cannot label a
    bf16x3 run f32.  Besides the rate against the datasheet peak (2.4 GHz) the object carries the shader clock the forward
    ran at (r3d_last_clock: cycle counter against the wall clock inside the kernel, of a free-running forward) and the rate
    against the peak AT that clock, and - from the committed PMC passes of the same workload - the HBM traffic per launch,
    the MFMA-busy fraction and the executed (as opposed to algorithmic) FLOP rate."""
    agg = {}
    dev = x.device
    fn = fn or (lambda: lifter(x, p))
    batch = batch if batch is not None else x.shape[0]
    key = key or ("b%d" % batch)
    prec = lifter.precision(dev)
    peak = PEAK_FP32_MFMA_TFLOPS if prec == "f32" else PEAK_BF16X3_TFLOPS
    # the clock of a BUSY chip: after a synchronisation the clocks need tens of forwards to come back (a reading behind three
    # forwards says 2.1 GHz where the steady state runs at 2.4), and the event-bracketed launches below run at yet another one
    for _ in range(40):
        fn()
    clk = lifter.last_clock_ghz(dev)
    lifter.profile_call(fn, dev)
    pair_ms = []
    for _ in range(reps):
        for r in lifter.profile_call(fn, dev):
            if r["kernel"] == "r3d_event_pair":        # the empty bracket: what the two event records cost
                pair_ms.append(r["ms"])
                continue
            a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
    pair = sorted(pair_ms)[len(pair_ms) // 2] if pair_ms else 0.0
    if os.environ.get("R3D_DUMP_LAUNCHES"):
        for r in lifter.profile_call(fn, dev):
            print("launch %2d %-20s blocks %5d  %8.1f us  %7.2f GFLOP  %6.1f TFLOP/s" % (
                r["stage"], r["kernel"], r["blocks"], r["ms"] * 1e3, r["flops"] / 1e9,
                r["flops"] / max(r["ms"], 1e-9) / 1e9), file=sys.stderr)
    # a bracket = the kernel + the event records around it; the empty bracket measures the latter (median of the
    # reps) and is taken off every launch
    for d in agg.values():
        d["raw_ms"] = d["ms"]
        d["ms"] = max(d["ms"] - pair * d["launches"], 0.5 * d["ms"])
    sum_ms = sum(d["ms"] for d in agg.values()) / reps                 # all kernels of one profiled step
    # Bracketed launches run slower than the free-running step (clock give-back, serialised event records); the
    # kernels of a step cannot take longer than the step, so the profile supplies the SHARES and the timed region the
    # total.  Never scaled up: idle gaps inside a step are not kernel time.
    scale = min(1.0, step_ms / sum_ms) if sum_ms > 0 else 1.0
    name = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[name]
    ms = d["ms"] * scale
    achieved = d["flops"] / (ms * 1e-3) / 1e12
    pmc = pmc_figures(key, name)
    launch_s = ms / d["launches"] * 1e-3
    out = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
           "peak_of": "fp32 MFMA (v_mfma_f32_32x32x2_f32)" if prec == "f32" else "bf16 MFMA / 6 products (bf16x3)",
           "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           "traffic": pmc["traffic_bytes"] if pmc else None,
           # the clock the kernel ran at (live) and what the peak is at that clock: the datasheet peak assumes 2.4 GHz
           "clk_ghz": round(clk, 3) if clk > 0 else None,
           "peak_at_clk": round(peak * clk / 2.4, 1) if clk > 0 else None,
           "frac_at_clk": round(achieved / (peak * clk / 2.4), 4) if clk > 0 else None,
           # from the committed PMC passes of this workload (profiles/pmc.json): SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles
           # of the launch, and the FLOPs those busy cycles executed over this run's launch time
           "mfma_busy_frac": pmc["mfma_busy_frac"] if pmc else None,
           "executed_tflops": round(pmc["executed_flops"] / launch_s / 1e12, 2) if pmc else None,
           # ... and that rate against the same peak: what the matrix cores really did, beside `frac`, which credits the
           # reference's arithmetic (a clip call evaluates expand_conv per frame and is credited per window; the folded first
           # layers execute fewer FLOPs than the reference's cat(x, diff, diff_t) form)
           "executed_frac": round(pmc["executed_flops"] / launch_s / 1e12 / peak, 4) if pmc else None,
           "pmc_source": "committed profile (profiles/pmc.json: rocprofv3 --pmc passes of this workload on an earlier run), not this run" if pmc else None,
           "pmc": {"table": "profiles/pmc.json[%s][%s]" % (key, name), "clk_ghz_pmc_pass": pmc["clk_ghz_pmc_pass"],
                   "l2_hit_pct": pmc["l2_hit_pct"], "fetch_bytes": pmc["fetch_bytes"], "write_bytes": pmc["write_bytes"]} if pmc else None,
           "launches_per_step": d["launches"] // reps,
           "avg_launch_us": round(ms / d["launches"] * 1e3, 2),
           "avg_launch_us_bracketed": round(d["ms"] / d["launches"] * 1e3, 2),
           "event_pair_us": round(pair * 1e3, 2), "scale_to_step": round(scale, 4),
           "step_us": round(step_ms * 1e3, 1),
           "kernels_us_per_step": {k: round(v["ms"] * scale / reps * 1e3, 1) for k, v in agg.items()},
           "flops_per_launch": d["flops"] / d["launches"],
           "hbm_view": {"algorithmic_GBps": round(d["bytes"] / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                        "frac": round(d["bytes"] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}}
    # which roof binds: the matrix time of the algorithmic FLOPs at peak against one stream of the algorithmic bytes at
    # 8 TB/s - a call of a few windows streams 200 MB of weights for a handful of FLOPs and is HBM-bound
    if d["bytes"] / (PEAK_HBM_GBS * 1e9) > d["flops"] / (peak * 1e12):
        hv = out["hbm_view"]
        out.update({"bound": "hbm", "achieved": hv["algorithmic_GBps"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hv["frac"],
                    "peak_of": "HBM3E", "mfma_view": {"achieved_TFLOPs": round(achieved, 2), "frac": round(achieved / peak, 4)}})
    return out


def bf16x3_line(dev, states, x, p, out_f32, args, barrier, cfg):
    """The windows workload again through a second pair of handles created with r3d_config.bf16x3 = 1
    (model_config['BF16X3'])."""
    import ray3d_amd
    from ray3d_amd import synth
    if os.environ.get("R3D_BF16X3") == "0":
        return {"skipped": "R3D_BF16X3=0 in the environment overrides the configuration"}
    lifter3, _ = build(dev, bf16x3=True)
    with torch.no_grad():
        lifter3.prepare([x.shape[0]], dev)
        o3 = lifter3(x, p)
        torch.cuda.synchronize()
        settle_clocks(lambda: lifter3(x, p), dev)
        el, dev_s, _ = timed_steps(lambda: lifter3(x, p), args.steps, args.warmup, barrier, dev)
        res = {"dtype": "bf16x3 (fp32-equivalent: fp32 operands split exactly into three bf16 terms, six products, fp32 accumulate)",
               "value": round(x.shape[0] * args.steps / el, 1), "unit": "poses/s", "ms_per_step": round(el / args.steps * 1e3, 4),
               "max_abs_diff_vs_f32_path_m": float((o3 - out_f32).abs().max().item()),
               "fp32_equivalent_TFLOPs": None}
        rl = roofline(lifter3, x, p, dev_s / args.steps * 1e3)
        res["fp32_equivalent_TFLOPs"] = rl["achieved"]
        res["frac_of_fp32_mfma_peak"] = round(rl["achieved"] / PEAK_FP32_MFMA_TFLOPS, 4)
        res["frac_of_bf16x3_peak"] = rl["frac"]                    # 2.5 PFLOP/s dense bf16, six products per fp32 product
        if x.shape[0] == BATCH and not args.no_b1024:
            xb = torch.from_numpy(synth.synth_rays(1024, cfg, seed=100)).to(dev)
            pb = torch.from_numpy(synth.synth_param(1024, seed=0, vary=False)).to(dev)
            lifter3.prepare([1024], dev)
            lifter3(xb, pb)
            nb = max(args.steps // 2, 5)
            settle_clocks(lambda: lifter3(xb, pb), dev, group=5, max_groups=20)
            elb, _, _ = timed_steps(lambda: lifter3(xb, pb), nb, max(args.warmup // 2, 2), barrier, dev)
            res["b1024"] = {"value": round(1024 * nb / elb, 1), "ms_per_step": round(elb / nb * 1e3, 4)}
    del lifter3
    return res


def half_chip_variant(dev, x, p, out_ref, steps, warmup):
    """EXPERIMENT, secondary object only (never `value` / `ms_per_step`): two persistent forwards side by side, each on a
    CU-masked stream of 128 CUs (ray3d_amd.masked_stream: both halves span all eight XCDs) with its own pair of handles
    (R3D_OPT_CU_LIMIT = 128: 128 workgroups per forward, no cross-stream ordering).  A round = one 256-window forward on
    EACH stream; the two do not depend on each other, so this is the throughput of back-to-back independent batches, not
    the latency of one.  At 256 windows the M = B levels of one forward leave 30 - 60 CUs idle (192 - 224 tiles for 256
    CUs); on 128 CUs the same levels run as two full rounds."""
    import ray3d_amd
    streams = [ray3d_amd.masked_stream(range(0, 128), dev), ray3d_amd.masked_stream(range(128, 256), dev)]
    lifters = []
    for _ in streams:
        l, _s = build(dev)
        l.set_cu_limit(128)
        lifters.append(l)
    B = x.shape[0]
    xs, ps = [x.clone() for _ in streams], [p.clone() for _ in streams]
    outs = [None, None]

    def round_():
        for i, (l, st) in enumerate(zip(lifters, streams)):
            with torch.cuda.stream(st):
                outs[i] = l(xs[i], ps[i])

    with torch.no_grad():
        for l, st in zip(lifters, streams):
            with torch.cuda.stream(st):
                l.prepare([B], dev)
        round_()
        torch.cuda.synchronize()
        for l in lifters:
            l.check_status(dev)
        err = max(float((o - out_ref).abs().max().item()) for o in outs)
        for _ in range(max(warmup, 30)):
            round_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            round_()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    del lifters
    return {"value": round(2 * B * steps / el, 1), "unit": "poses/s", "ms_per_round": round(el / steps * 1e3, 4),
            "rounds": steps, "windows_per_round": 2 * B, "max_abs_diff_vs_whole_chip_path_m": err,
            "note": "EXPERIMENT: two independent %d-window forwards per round, one per CU-masked stream of 128 CUs (two pairs of "
                    "handles, R3D_OPT_CU_LIMIT); throughput of back-to-back independent batches - not the headline's step" % B}


def lanes_variant(lifter, dev, x, p, out_ref, steps, warmup, n):
    """SECONDARY object (never `value`): R3D_OPT_LANES = n on the run's OWN pair of handles - n library-owned CU-masked streams that
    share one packed weight image - lifting n independent batches per round side by side.  What round 5 measured with two model
    copies and caller-made streams (`half_chip_streams_variant`), through one pair.  Reports the device memory the lanes added."""
    B = x.shape[0]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    lifter.set_lanes(n, dev)
    xs, ps = [x.clone() for _ in range(n)], [p.clone() for _ in range(n)]
    outs = [None] * n

    def round_():
        for k in range(n):
            with lifter.lane(k):
                outs[k] = lifter(xs[k], ps[k])

    try:
        with torch.no_grad():
            lifter.prepare([B], dev)
            round_()
            lifter.join_lanes()
            torch.cuda.synchronize()
            lifter.check_status(dev)
            added = free0 - torch.cuda.mem_get_info(dev)[0]
            err = max(float((o - out_ref).abs().max().item()) for o in outs)
            for _ in range(max(warmup, 30)):
                round_()
            lifter.join_lanes()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                round_()
            lifter.join_lanes()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    finally:
        torch.cuda.synchronize()
        lifter.set_lanes(0, dev)
    from ray3d_amd import _capi
    ws_bytes = _capi.workspace_bytes(lifter.pos.handle(dev), lifter.trj.handle(dev), B)
    return {"lanes": n, "value": round(n * B * steps / el, 1), "unit": "poses/s", "ms_per_round": round(el / steps * 1e3, 4),
            "rounds": steps, "windows_per_round": n * B, "max_abs_diff_vs_whole_chip_path_m": err,
            "device_bytes_added_by_lanes": int(added), "workspace_bytes_per_lane": int(ws_bytes),
            "device_bytes_added_minus_workspaces_and_inputs": int(added - n * ws_bytes - sum(t.numel() * 4 for t in xs + ps) - n * B * 51 * 4),
            "note": "SECONDARY: %d independent %d-window forwards per round on %d library-owned lanes (R3D_OPT_LANES) of ONE pair of handles - "
                    "one weight image; throughput of back-to-back independent batches, not the headline's step" % (n, B, n)}


def c1024_line(dev, steps, warmup, barrier, batches=(256,)):
    """SURVEY 8 f4 / lib/model/rie.py:14: TemporalBlock's class default is channels = 1024 (no shipped cfg uses it: every cfg file
    sets 256).  Such models run the LEVEL-BY-LEVEL form - the single persistent launch holds tiles of at most 256 channels - so
    this object carries what that form costs: poses/s, the roofline of its dominant kernel (r3d_gemm_f32, twelve launches per
    step), and the gaps between its launches (step time minus the sum of the kernels' own times)."""
    import ray3d_amd
    from ray3d_amd import synth
    lifter, states = build(dev, CHANNELS=1024)
    cfg = states["pos"][0]
    res = {"workload": "J 17, RF 243, CHANNELS = 1024 (the reference's class default, rie.py:14), latent 256, stage 3, pos + trj; "
                       "level-by-level form (models of more than 256 channels do not run as one persistent launch)",
           "params_M": round(sum(float(np.asarray(v).size) for st in (states["pos"][1], states["trj"][1]) for v in st.values()) / 1e6, 1),
           "parity": "tests/test_gpu_parity.py::test_other_widths_match_oracle (C = 512 / 1024 against the oracle at the literal 1e-4 bound)"}
    with torch.no_grad():
        for B in batches:
            x = torch.from_numpy(synth.synth_rays(B, cfg, seed=100)).to(dev)
            p = torch.from_numpy(synth.synth_param(B, seed=0, vary=False)).to(dev)
            lifter.prepare([B], dev)
            run = lambda: lifter(x, p)
            assert torch.isfinite(run()).all()
            settle_clocks(run, dev, group=5, max_groups=20)
            el, dev_s, _ = timed_steps(run, steps, warmup, barrier, dev)
            rl = roofline(lifter, x, p, dev_s / steps * 1e3, batch=B, key="c1024_b%d" % B)
            kern = sum(rl["kernels_us_per_step"].values())
            res["b%d" % B] = {"value": round(B * steps / el, 1), "unit": "poses/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
                              "roofline": rl,
                              "launches_per_step": sum(1 for _ in lifter.profile_call(run, dev)) - 1,
                              # bracketed launches scaled to the step: what is left of the step is launch gaps (scale_to_step < 1 means the
                              # bracketed kernels alone already exceed the free-running step: no gap can be read off then)
                              "kernels_us_per_step_sum": round(kern, 1),
                              "gaps_us_per_step": round(max(rl["step_us"] - kern, 0.0), 1) if rl["scale_to_step"] >= 1.0 else None}
    del lifter
    return res


def parity_gate(lifter, dev, batch):
    """no timing counts before parity.  The reference's own outputs for the weights this bench
    builds (seeds 1 / 2, decoder scale 1: tests/golden/model_j17_rf243_s3.npz, written by the reference's PyTorch-CPU
    forward) against the HIP path - the fixture's windows tiled to the TIMED batch size, so that it is the timed launch
    plan and tile schedule that is checked.  Returns (max abs error, max |reference|); exits above PARITY_ATOL."""
    z = np.load(PARITY_FIXTURE)
    reps = -(-batch // z["x"].shape[0])
    x = np.tile(z["x"], (reps, 1, 1, 1))[:batch]
    p = np.tile(z["param"], (reps, 1))[:batch]
    ref = np.tile(z["out_pos"] + z["out_trj"], (reps, 1, 1, 1))[:batch]
    with torch.no_grad():
        out = lifter(torch.from_numpy(x).to(dev), torch.from_numpy(p).to(dev)).cpu().numpy()
    err = float(np.abs(out.astype(np.float64) - ref).max())
    if not np.isfinite(out).all() or err > PARITY_ATOL:
        raise SystemExit("bench.py: parity gate failed: max abs error %.3e vs the reference fixture (bound %.0e) - nothing "
                         "is timed on a path whose results differ from the reference's" % (err, PARITY_ATOL))
    return err, float(np.abs(ref).max())


def dhp_cameras():
    """The 14 MPI-INF-3DHP cameras (lib/dataset/mpii_3dhp_dataset.py:9-251) from the reference-generated fixture."""
    import ray3d_amd
    z = np.load(os.path.join(ROOT, "tests", "golden", "cameras_3dhp.npz"))
    return [ray3d_amd.Camera(z[t + "/K"], z[t + "/R"], z[t + "/t"], name=str(t)) for t in z["tags"]]


def grid_cameras():
    """The 342 cameras of the 'Train' augmentation grid (data/camera_augmentation.py:637-642) around H36M S1 / camera 1."""
    import ray3d_amd
    z = np.load(os.path.join(ROOT, "tests", "golden", "frontends.npz"))
    K = np.array([[1145.51133842, 0, 514.968197319], [0, 1144.77392808, 501.882018537], [0, 0, 1.0]])
    return ray3d_amd.camera_grid(K, z["grid/R0"], z["grid/T0"])


WORKLOADS = {
    # name: (ARCHITECTURE, joints, windows per step over all ranks, camera set, description)
    "cfg4_rf9": ("3,3", 17, 1024, "3dhp",
                 "BASELINE configs[3] as shipped (cfg_ray3d_3dhp_stage3.py:77-89: ARCHITECTURE '3,3', 9 frames, C=256): 17 joints, "
                 "pixel keypoints in, one of the 14 MPI-INF-3DHP cameras per window, rays encoded in the first-level gather"),
    "cfg4_rf243": ("3,3,3,3,3", 17, 1024, "3dhp",
                   "BASELINE configs[3] at RF 243: 17 joints, pixel keypoints in, one of the 14 MPI-INF-3DHP cameras per window"),
    "cfg5": ("3,3", 14, 4096, "grid",
             "BASELINE configs[4]: 14-joint layout, RF 9, 4096 windows per step over all ranks, pixel keypoints in, one camera of the "
             "342-camera augmentation grid (data/camera_augmentation.py:637-642) per window"),
}


def uv_workload(name, dev, world, rank, steps, warmup, barrier, want_rays_delta=True):
    """One of WORKLOADS on this rank: build the pair, synthesise pixel windows + per-window cameras, check against the
    rays mode (bit-identical by construction: the same float64 encode), time UV mode and rays mode, roofline of UV mode."""
    from ray3d_amd import synth
    arch, J, total, camset, desc = WORKLOADS[name]
    B = max(total // world, 1)
    lifter, states = build(dev, arch=arch, NUM_KPTS=J)
    cfg = states["pos"][0]
    rf = cfg.receptive_field
    cams = dhp_cameras() if camset == "3dhp" else grid_cameras()
    pick = [(5 * i + i // len(cams) + rank) % len(cams) for i in range(B)]
    if camset == "3dhp":
        uv = (2048.0 * synth.hash_uniform("bench.uv.%s.%d" % (name, rank), (B, rf, J, 2), 17)).astype(np.float32)
    else:
        rng = np.random.default_rng([14, rank])
        w = rng.normal(0, 0.25, (B, 1, J, 3)) + np.array([0, 0, 1.0]) + 0.01 * np.cumsum(rng.normal(0, 1, (B, rf, 1, 3)), axis=1)
        uv = np.stack([cams[c].project(w[i]) for i, c in enumerate(pick)]).astype(np.float32)
    rows = np.stack([cams[c].cam_row() for c in pick])
    par = np.stack([cams[c].param() for c in pick])
    uvd, rowsd, pard = torch.from_numpy(uv).to(dev), torch.from_numpy(rows).to(dev), torch.from_numpy(par).to(dev)
    run_uv = lambda: lifter.forward_uv(uvd, rowsd, pard)
    res = {"workload": desc, "batch_per_gpu": B, "receptive_field": rf, "joints": J, "cameras": len(cams),
           "input": "uv (pixels) + per-window camera rows"}
    with torch.no_grad():
        lifter.prepare([B], dev)
        out = run_uv()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        settle_clocks(run_uv, dev, group=10, max_groups=30)
        el, dev_s, _ = timed_steps(run_uv, steps, warmup, barrier, dev)
        res.update({"value": round(world * B * steps / el, 1), "unit": "poses/s", "ms_per_step": round(el / steps * 1e3, 4),
                    "steps": steps})
        rl = roofline(lifter, uvd, pard, dev_s / steps * 1e3, fn=run_uv, batch=B, key=name)
        flops_step = rl["flops_per_launch"] * rl["launches_per_step"]
        # which roof binds: the GEMMs have M = B rows (MLPs are 95 % of the FLOPs at RF 9) - matrix-bound when the
        # weights' HBM stream (once per step) is shorter than the matrix time, weight-bandwidth-bound at small B
        wbytes = sum(float(np.asarray(v).size) * 4 for st in (states["pos"][1], states["trj"][1]) for k, v in st.items()
                     if k.endswith("weight") and np.asarray(v).ndim >= 2)
        rl["binding_roof"] = {"mfma_us_at_peak": round(flops_step / (rl["peak"] * 1e12) * 1e6, 1),
                              "weight_stream_us_at_8TBps": round(wbytes / (PEAK_HBM_GBS * 1e9) * 1e6, 1),
                              "binds": "mfma" if flops_step / (rl["peak"] * 1e12) > wbytes / (PEAK_HBM_GBS * 1e9) else "hbm (weights)"}
        res["roofline"] = rl
        if want_rays_delta:
            rays = np.stack([cams[c].rays_from_uv(uv[i].astype(np.float64)) for i, c in enumerate(pick)]).astype(np.float32)
            rd = torch.from_numpy(rays).to(dev)
            run_rays = lambda: lifter(rd, pard)
            o2 = run_rays()
            res["uv_equals_rays_mode"] = bool(torch.equal(out, o2))
            settle_clocks(run_rays, dev, group=10, max_groups=10)
            el2, _, _ = timed_steps(run_rays, steps, warmup, barrier, dev)
            res["rays_mode"] = {"value": round(world * B * steps / el2, 1), "ms_per_step": round(el2 / steps * 1e3, 4)}
            res["uv_vs_rays_ms_delta"] = round((el - el2) / steps * 1e3, 4)
    del lifter
    return res


def synthetic_eval_set(n_clips, seed=0):
    """Clip lengths and cameras of the H36M-shaped stand-in: a pure function of (n_clips, seed), so every rank derives
    the same set (and the same sharding) without communicating.  Clip i is generated from its own stream."""
    import ray3d_amd
    rng = np.random.default_rng(seed)
    lengths = [int(rng.integers(1000, 6001)) for _ in range(n_clips)]
    cams = [ray3d_amd.synthetic_camera(yaw, 4.5, -12.0, name="cam%d" % i) for i, yaw in enumerate((20, 110, 200, 290))]
    return lengths, cams


def make_clip(i, n, cams, seed=0):
    from ray3d_amd import evaluate
    rng = np.random.default_rng([seed, i])
    cam = cams[i % 4]
    world = rng.normal(0, 0.3, (1, 17, 3)) + np.array([0, 0, 1.0]) + 0.02 * np.cumsum(rng.normal(0, 1.0, (n, 1, 3)), axis=0) \
        + rng.normal(0, 0.02, (n, 17, 3))
    rays = cam.rays_from_uv(cam.project(world)).astype(np.float32)
    return evaluate.Clip(cam, rays, cam.world2normalized(world).astype(np.float32), "A%d" % (i % 15), i)


def eval_partition(n_clips, world, seed=0, length_div=1):
    """Everything about the clip-sharded evaluation that is a pure function of (n_clips, world, seed) - so that every
    rank derives it without communicating: clip lengths and cameras, the whole-clip shards (longest first), the action ids.
    `length_div` shortens the clips (CPU tests of this very path: tests/test_host.py, world size 8 over gloo)."""
    from ray3d_amd import evaluate
    lengths, cams = synthetic_eval_set(n_clips, seed)
    lengths = [max(n // length_div, 2) for n in lengths]
    actions = sorted(set("A%d" % (i % 15) for i in range(n_clips)))
    return {"lengths": lengths, "cams": cams, "shards": evaluate.shard_clips(lengths, world),
            "aid": {a: i for i, a in enumerate(actions)}}


def eval_summary(allrows):
    """The gathered per-clip rows -> what the line reports: the action-wise averages (trainer.py:473-477) and a checksum of
    the MPJPE column in clip order.  Neither depends on the number of ranks."""
    from ray3d_amd import evaluate
    avg = evaluate.action_average(evaluate.reduce_partials(allrows))
    return {"action_average": avg[0], "p_mpjpe": avg[1], "n_mpjpe": avg[2], "mpjve": avg[3], "mrpe": avg[4],
            "checksum": float(allrows[allrows[:, 0].argsort()][:, 3].sum().item())}


def eval_pass(lifter, dev, dist, world, rank, n_clips, steps, warmup, barrier, max_over_ranks, all_ranks, cross_check=True,
              length_div=1, lift=None, lanes=0):
    """BASELINE configs[2] / north_star's multi-GPU split on this process group: the synthetic clip set sharded over the ranks as
    whole clips (longest first), every clip lifted with in-kernel sliding windows and reduced to one row on the device, ONE
    all_gather of the rows per pass (lib/train_val/trainer.py:399-403,473-477 reduce them per action).  Strong scaling: the
    set is fixed.  Returns (on every rank) a dict; the fields that need the gathered rows are complete on rank 0.
    `cross_check`: every rank also lifts the first clip of the NEXT rank's shard, and the row it computes must equal the
    gathered one bit for bit - what a clip's row is does not depend on which rank (or how many ranks) lifted it.
    `lift(clip, out_row)` replaces the HIP path (forward_clip + r3d_clip_metrics) and `length_div` shortens the clips: the CPU
    test of this function's own sharding / rows / gather / cross-check / summary logic (tests/test_host.py, world size 8 over gloo)."""
    from ray3d_amd import evaluate
    dev = torch.device(dev)
    on_gpu = dev.type == "cuda"
    part = eval_partition(n_clips, world, length_div=length_div)
    lengths, cams, shards, aid = part["lengths"], part["cams"], part["shards"], part["aid"]

    def resident(idx):
        c = make_clip(idx, lengths[idx], cams)
        if lift is not None:
            return (c, None, None, None)
        padded = torch.from_numpy(evaluate.pad_clip(c.rays, 121)).to(dev)
        return (c, padded, torch.from_numpy(c.camera.param()).to(dev), torch.from_numpy(c.gt_norm).to(dev))

    def lift_row(item, out_row):
        c, padded, prow, gt = item
        if lift is not None:
            lift(c, out_row)
        elif lanes:
            # R3D_OPT_LANES: the clip's forward AND its metrics on the next lane's stream (clips are independent: `lanes` of them share the chip)
            with lifter.lane():
                evaluate.clip_partials_hip(lifter.forward_clip(padded, prow), c, aid[c.action], gt_dev=gt, out=out_row)
        else:
            evaluate.clip_partials_hip(lifter.forward_clip(padded, prow), c, aid[c.action], gt_dev=gt, out=out_row)

    t0 = time.perf_counter()
    mine = [resident(idx) for idx in shards[rank]]
    sizes = None
    if lift is None:
        sizes = sorted(set(b for c, _, _, _ in mine for b in lifter.clip_batch_sizes(c.rays.shape[0])))
        lifter.prepare(sizes, dev)
    setup_s = time.perf_counter() - t0
    # the rank's per-clip rows: header columns (clip, action, frames) uploaded ONCE, error columns written on the device
    local_rows = evaluate.partial_rows([(c.clip_id, aid[c.action], c.rays.shape[0]) for c, _, _, _ in mine], dev)
    counts = [len(s_) for s_ in shards]

    def lift_mine():
        for k, item in enumerate(mine):
            lift_row(item, local_rows[k])
        if lanes:
            lifter.join_lanes()

    def one_pass():
        lift_mine()
        return evaluate.gather_partials(local_rows, counts) if dist is not None else local_rows

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    with torch.no_grad():
        one_pass()                          # first touch: workspace allocation
        elapsed_own, dev_s, allrows = timed_steps(one_pass, steps, warmup, barrier, dev)
        # this rank's own pass (its clips, no gather): what the shard costs without waiting for the others
        barrier()
        t1 = time.perf_counter()
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        lift_mine()
        if on_gpu:
            e1.record(torch.cuda.current_stream(dev))
            e1.synchronize()
            own_pass_ms = e0.elapsed_time(e1)             # device time
        else:
            own_pass_ms = (time.perf_counter() - t1) * 1e3
        # ... and the single exchange step by itself, behind a barrier (host wall clock: the collective runs on RCCL's stream)
        barrier()
        tg = time.perf_counter()
        if dist is not None:
            evaluate.gather_partials(local_rows, counts)
        sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
        cross_ok = None
        if cross_check and world > 1 and shards[(rank + 1) % world]:
            idx = shards[(rank + 1) % world][0]
            item = resident(idx)
            c = item[0]
            row = evaluate.partial_rows([(c.clip_id, aid[c.action], c.rays.shape[0])], dev)
            lift_row(item, row[0])
            sync()
            theirs = allrows[allrows[:, 0] == float(idx)]
            cross_ok = float(theirs.shape[0] == 1 and bool(torch.equal(theirs[0], row[0])))
    elapsed = max_over_ranks(elapsed_own)
    per_rank_pass_ms = all_ranks(own_pass_ms)
    gather_ms = max_over_ranks(gather_ms)
    cross = all_ranks(cross_ok) if cross_ok is not None else None
    shard_frames = [sum(lengths[i] for i in sh) for sh in shards]
    frames = sum(lengths)
    res = {"value": round(frames * steps / elapsed, 1), "unit": "poses/s", "ms_per_pass": round(elapsed / steps * 1e3, 3),
           "passes": steps, "warmup_passes": warmup, "scaling": "strong",
           "clips": n_clips, "frames": frames, "receptive_field": 243, "joints": 17,
           "batch_sizes": sizes if world == 1 else None,
           "world_size_observed": world if dist is None else dist.get_world_size(),
           "backend": None if dist is None else dist.get_backend(),
           "shard_frames": shard_frames,
           "shard_imbalance": round(max(shard_frames) / (sum(shard_frames) / world), 4),   # max / mean frames per rank
           "pass_ms_per_rank": [round(v, 3) for v in per_rank_pass_ms],                    # device time of each rank's own clips
           "pass_ms_imbalance": round(max(per_rank_pass_ms) / (sum(per_rank_pass_ms) / world), 4),
           "all_gather_ms": round(gather_ms, 3),
           "all_gather": "ONE all_gather of %d x %d float64 per pass (%d bytes per rank after padding to the largest shard)"
                         % (n_clips, evaluate.PARTIAL_COLS, max(max(counts), 1) * evaluate.PARTIAL_COLS * 8),
           "cross_rank_rows_bit_equal": None if cross is None else bool(all(v == 1.0 for v in cross)),
           "setup_s": round(setup_s, 2)}
    if rank == 0:
        assert allrows.shape[0] == n_clips and sorted(int(v) for v in allrows[:, 0].tolist()) == list(range(n_clips))
        res["mpjpe_mm"] = eval_summary(allrows)       # action-wise averages + checksum: neither depends on the number of ranks
        # every window is 251.7 MFLOP of the reference's arithmetic, whatever the clip path shares between windows
        res["algorithmic_tflops"] = round(frames * steps / elapsed * 251.7e6 / 1e12, 2)
        res["frac_of_fp32_mfma_peak_all_gpus"] = round(res["algorithmic_tflops"] / (PEAK_FP32_MFMA_TFLOPS * world), 4)
    res["_mine"] = mine
    return res


def self_launch(args):
    """--gpus N without a launcher: start N ranks through torch.distributed.run, forward their output."""
    n = torch.cuda.device_count()
    if n < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible to this process" % (args.gpus, n))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=("windows", "eval"), default="windows")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("cfg2",) + tuple(WORKLOADS), default="cfg2",
                    help="windows mode: cfg2 = BASELINE configs[1] (the headline); cfg4_* / cfg5 = the shipped RF-9 configurations")
    ap.add_argument("--no-shipped-cfgs", action="store_true", help="windows mode: skip the secondary cfg4 / cfg5 objects")
    ap.add_argument("--clips", type=int, default=EVAL_CLIPS, help="eval mode: number of clips in the set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 2, 4),
                    help="also measure R3D_OPT_LANES = N on the run's own handles: windows mode `lanes_variant` (N independent batches per round), "
                         "eval mode `lanes` (clips on N lanes; same MPJPE)")
    ap.add_argument("--no-eval-pass", action="store_true",
                    help="windows mode under a process group: skip the secondary clip-sharded evaluation pass (`eval_pass`)")
    ap.add_argument("--no-b1024", action="store_true", help="windows mode: skip the roofline point at 1024 windows")
    ap.add_argument("--no-c1024", action="store_true", help="windows mode: skip the secondary object of the 1024-channel model (level-by-level form)")
    ap.add_argument("--c1024", action="store_true", help="only the 1024-channel object, at 64 and 256 windows, as its own JSON line (profile recipes)")
    ap.add_argument("--no-bf16x3", action="store_true", help="windows mode: skip the secondary bf16x3 (fp32-equivalent) measurement")
    ap.add_argument("--two-stream", action="store_true", help="also time Ray3DLifter.forward_overlapped (two half batches on two streams)")
    ap.add_argument("--half-chip-streams", action="store_true",
                    help="also time two independent forwards side by side on two CU-masked streams of 128 CUs (secondary object `half_chip_streams_variant`; `two_stream_variant` keeps its round-4 meaning: --two-stream)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 50 if args.mode == "windows" else 3
    if args.warmup is None:
        args.warmup = 10 if args.mode == "windows" else 1

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with torch.distributed.run "
                         "--nproc-per-node %d (or without a launcher: it starts its own ranks)" % (args.gpus, world, args.gpus))
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        # (under a launcher the collectives run even with one rank: the same code path as N > 1, over RCCL)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == world

    bar_token = torch.zeros(1, dtype=torch.float32, device=dev) if dist is not None else None

    def barrier():
        # the device's work done, then a collective that every rank has to enter (a one-element all_reduce over RCCL on a
        # preallocated tensor), then its completion: a barrier + synchronize.  (dist.barrier() does the same through a blocking
        # wait of its own whose wake-up granularity - 0 .. 0.7 ms per call, measured under the launcher with one rank - would be
        # charged to the K timed steps: 20 steps are 11 ms.)
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(bar_token)
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(v):
        """[v of rank 0, v of rank 1, ...] on every rank."""
        if dist is None:
            return [v]
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        bucket = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bucket, t)
        return [float(b.item()) for b in bucket]

    from ray3d_amd import evaluate, synth
    if args.mode == "windows" and args.workload != "cfg2":
        # ---- one of the shipped configurations as the run's workload (rocprofv3 recipes use this form)
        arch, J, total, camset, desc = WORKLOADS[args.workload]
        res = uv_workload(args.workload, dev, world, rank, args.steps, args.warmup, barrier)
        el = max_over_ranks(res["ms_per_step"])
        per_rank = all_ranks(res["ms_per_step"])
        if rank == 0:
            B = res["batch_per_gpu"]
            print(json.dumps({
                "metric": "lifted poses/sec (%d-joint, %d-frame window)" % (J, res["receptive_field"]), "unit": "poses/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
                "dtype": "f32" if "fp32" in res["roofline"]["peak_of"] else "bf16x3",
                "data": "synthetic", "value": round(world * B / (el * 1e-3), 1), "ms_per_step": el, "scaling": "weak",
                "config": {"workload": desc, "batch_per_gpu": B, "receptive_field": res["receptive_field"], "joints": J,
                           "parallelism": "dp%d (independent window batches, no data-path collective)" % world,
                           "world_size_observed": world if dist is None else dist.get_world_size(),
                           "ms_per_step_per_rank": per_rank},
                "roofline": res["roofline"], "rays_mode": res.get("rays_mode"), "uv_vs_rays_ms_delta": res.get("uv_vs_rays_ms_delta"),
                "uv_equals_rays_mode": res.get("uv_equals_rays_mode")}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == "windows" and args.c1024:
        res = c1024_line(dev, args.steps, args.warmup, barrier, batches=(64, 256))
        if rank == 0:
            print(json.dumps({"metric": "lifted poses/sec (17-joint, 243-frame window, CHANNELS 1024)", "unit": "poses/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "scaling": "weak", "value": res["b256"]["value"], "ms_per_step": res["b256"]["ms_per_step"],
                              "config": {"workload": res["workload"], "batch_per_gpu": 256}, "roofline": res["b256"]["roofline"], "c1024": res}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    lifter, states = build(dev)
    cfg = states["pos"][0]
    line = {"metric": "lifted poses/sec (17-joint, 243-frame window)", "unit": "poses/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": lifter.precision(dev),    # what the handles actually compute in (r3d_precision): f32 unless R3D_BF16X3=1 / BF16X3
            "data": "synthetic"}

    if args.mode == "windows":
        x_np = synth.synth_rays(args.batch, cfg, seed=100 + rank)
        p_np = synth.synth_param(args.batch, seed=0, vary=False)
        x, p = torch.from_numpy(x_np).to(dev), torch.from_numpy(p_np).to(dev)
        with torch.no_grad():
            # one-time set-up of the library for this batch size (launch plan, tile schedule upload, workspace
            # allocation) - not a warm-up step; its cost is reported
            lifter.pos.handle(dev), lifter.trj.handle(dev)      # weights folded, packed and uploaded (r3d_finalize)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lifter.prepare([args.batch], dev)                   # plan + tile schedule of this batch size, uploaded
            prepare_ms = (time.perf_counter() - t0) * 1e3
            lifter(x, p)
            torch.cuda.synchronize()
            # parity gate BEFORE the timed region (every rank: its own device runs its own copy of the weights)
            parity_err, parity_ref = parity_gate(lifter, dev, args.batch)
            # the literal command first - W warm-up steps, K timed steps, the GPU's clocks still ramping (a process starts in a
            # low power state) - reported beside the sustained number, never instead of it
            torch.cuda.synchronize()
            time.sleep(0.2)
            el_cold, _, _ = timed_steps(lambda: lifter(x, p), args.steps, args.warmup, barrier, dev)
            settle_steps = settle_clocks(lambda: lifter(x, p), dev)
            elapsed_own, dev_s, out = timed_steps(lambda: lifter(x, p), args.steps, args.warmup, barrier, dev)
        elapsed = max_over_ranks(elapsed_own)
        per_rank_ms = all_ranks(elapsed_own / args.steps * 1e3)
        parity_err = max_over_ranks(parity_err)
        assert torch.isfinite(out).all()
        # With a process group (the driver's N > 1 command; any run under a launcher): north_star's multi-GPU split as a
        # SECONDARY object of the same line - one strong-scaling pass set over the 240-clip evaluation shape, clips sharded
        # over the ranks, ONE all_gather of the per-clip rows per pass.  The windows-mode `value` above has no collective on
        # its data path; without this object an N-GPU run of this command would say nothing about clip sharding.
        ev = None
        if dist is not None and args.batch == BATCH and not args.no_eval_pass and lifter.precision(dev) == "f32":
            try:
                ev = eval_pass(lifter, dev, dist, world, rank, args.clips, 2, 1, barrier, max_over_ranks, all_ranks)
                ev.pop("_mine")
            except Exception as e:                          # noqa: BLE001 - reported in the line, never costs the headline
                ev = {"error": "%s: %s" % (type(e).__name__, e)}
                print("bench.py: eval_pass failed on rank %d: %r" % (rank, e), file=sys.stderr)
        if rank == 0:
            line.update({
                "value": round(world * args.batch * args.steps / elapsed, 1),
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "scaling": "weak",
                "config": {"workload": "BASELINE configs[1]: synthetic 17-joint 243-frame windows, batch %d per GPU, "
                                       "forward-only pos+trj (C=256, latent 256, stage 3, camera embedding)" % args.batch,
                           "batch_per_gpu": args.batch, "receptive_field": 243, "joints": 17,
                           "parallelism": "dp%d (independent window batches, no data-path collective)" % world,
                           "schedule_build_ms": round(prepare_ms, 1),
                           "clock_settle_steps": settle_steps,
                           "ms_per_step_before_clock_settle": round(el_cold / args.steps * 1e3, 4),   # the same K steps right after W warm-up steps from idle clocks
                           "world_size_observed": world if dist is None else dist.get_world_size(),
                           "ms_per_step_per_rank": [round(v, 4) for v in per_rank_ms]},
                "parity_max_abs_err": parity_err,
                "parity": {"max_abs_err_m": parity_err, "bound_m": PARITY_ATOL, "ref_max_abs_m": round(parity_ref, 3),
                           "against": "tests/golden/model_j17_rf243_s3.npz (reference PyTorch-CPU outputs for these weights), "
                                      "tiled to the timed batch size; checked before the timed region, on every rank (max)"}})
            # (the extras below must never cost the run its line: a failure in one of them is reported in its place)
            def guarded(key, fn):
                try:
                    line[key] = fn()
                except Exception as e:                      # noqa: BLE001 - reported, not swallowed
                    line[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                    print("bench.py: %s failed: %r" % (key, e), file=sys.stderr)

            def _roofline():
                with torch.no_grad():
                    return roofline(lifter, x, p, dev_s / args.steps * 1e3)
            guarded("roofline", _roofline)
            if world == 1 and args.batch == BATCH and not args.no_b1024:
                # north_star quotes its roofline target at 1024 x 243 x 17: the same measurement at that batch
                def _b1024():
                    xb = torch.from_numpy(synth.synth_rays(1024, cfg, seed=100)).to(dev)
                    pb = torch.from_numpy(synth.synth_param(1024, seed=0, vary=False)).to(dev)
                    with torch.no_grad():
                        lifter.prepare([1024], dev)
                        lifter(xb, pb)
                        settle_clocks(lambda: lifter(xb, pb), dev, group=5, max_groups=20)
                        el_b, dev_b, _ = timed_steps(lambda: lifter(xb, pb), max(args.steps // 2, 5), max(args.warmup // 2, 2), barrier, dev)
                        nb = max(args.steps // 2, 5)
                        rb = roofline(lifter, xb, pb, dev_b / nb * 1e3)
                    rb.update({"batch": 1024, "value": round(1024 * nb / el_b, 1), "ms_per_step": round(el_b / nb * 1e3, 4), "steps": nb})
                    return rb
                guarded("roofline_b1024", _b1024)
            if world == 1 and args.batch == BATCH and not args.no_shipped_cfgs:
                # the configurations the reference ships (every cfg file is RF 9): BASELINE configs[3] and configs[4]
                nsh = max(args.steps // 2, 10)
                guarded("cfg4", lambda: {"rf9": uv_workload("cfg4_rf9", dev, 1, 0, nsh, 3, barrier),
                                         "rf243": uv_workload("cfg4_rf243", dev, 1, 0, max(nsh // 2, 5), 2, barrier)})
                guarded("cfg5", lambda: uv_workload("cfg5", dev, 1, 0, nsh, 3, barrier))
            if world == 1 and args.batch == BATCH and not args.no_c1024:
                # the reference's class-default width (1024 channels): the level-by-level form's cost, with its roofline
                guarded("c1024", lambda: c1024_line(dev, max(args.steps // 4, 5), 2, barrier))
            if world == 1 and not args.no_bf16x3 and lifter.precision(dev) == "f32":
                # secondary line: the same workload with r3d_config.bf16x3 = 1 - every big GEMM on the bf16 matrix cores with exact
                # three-term splits of both operands (fp32-equivalent results: tests/test_gpu_parity.py holds it to the
                # fp32 path's own error against a float64 reference).  Not the headline: `dtype` above stays f32.
                guarded("bf16x3", lambda: bf16x3_line(dev, states, x, p, out, args, barrier, cfg))
            if world == 1 and args.two_stream:
                with torch.no_grad():
                    el2, _, _ = timed_steps(lambda: lifter.forward_overlapped(x, p), args.steps, args.warmup, barrier, dev)
                line["two_stream_variant"] = {"value": round(args.batch * args.steps / el2, 1), "unit": "poses/s",
                                              "ms_per_step": round(el2 / args.steps * 1e3, 4),
                                              "note": "same work as `value`, issued as 2 half batches on 2 streams"}
            if world == 1 and args.lanes > 1 and lifter.precision(dev) == "f32":
                guarded("lanes_variant", lambda: lanes_variant(lifter, dev, x, p, out, args.steps, args.warmup, args.lanes))
            if world == 1 and args.half_chip_streams and lifter.precision(dev) == "f32":
                guarded("half_chip_streams_variant", lambda: half_chip_variant(dev, x, p, out, args.steps, args.warmup))
            if ev is not None:
                ev["note"] = ("SECONDARY: BASELINE configs[2]'s shape on this run's process group - %d synthetic clips (%d frames) sharded over "
                              "%d rank(s) as whole clips, in-kernel sliding windows, device metrics, one all_gather of the per-clip rows per "
                              "pass; strong scaling (the set is fixed): compare `value` / `ms_per_pass` across N, `mpjpe_mm` must not change"
                              % (ev.get("clips", 0), ev.get("frames", 0), world)) if "error" not in ev else ev.get("error")
                line["eval_pass"] = ev
            if not args.no_cpu_baseline:
                # (rank 0 only, at every N: the reference's CPU path timed on this host beside the GPU numbers)
                guarded("cpu_baseline", lambda: cpu_baseline(states, x_np, p_np))
            print(json.dumps(line))
    else:
        # ---- clip-sharded evaluation (configs[2]): whole clips per rank, resident in HBM, one all_gather per pass
        ev = eval_pass(lifter, dev, dist, world, rank, args.clips, args.steps, args.warmup, barrier, max_over_ranks, all_ranks)
        mine = ev.pop("_mine")
        if rank == 0:
            frames = ev["frames"]
            line.update({
                "value": ev["value"], "ms_per_step": ev["ms_per_pass"], "scaling": "strong",
                "config": {"workload": "BASELINE configs[2] stand-in: %d synthetic clips of U(1000,6000) frames (%d frames), 17 joints, "
                                       "RF 243, whole clips sharded over the ranks longest-first, in-kernel sliding windows, device "
                                       "metrics, one RCCL all_gather of the per-clip partial rows per pass" % (args.clips, frames),
                           "clips": args.clips, "frames": frames, "receptive_field": 243, "joints": 17,
                           "batch_sizes": ev["batch_sizes"],
                           "world_size_observed": ev["world_size_observed"],
                           "shard_frames": ev["shard_frames"], "shard_imbalance": ev["shard_imbalance"],
                           "pass_ms_per_rank": ev["pass_ms_per_rank"], "pass_ms_imbalance": ev["pass_ms_imbalance"],
                           "all_gather_ms": ev["all_gather_ms"], "cross_rank_rows_bit_equal": ev["cross_rank_rows_bit_equal"],
                           "parallelism": "clips sharded over %d rank(s); collective = %s" % (world, ev["all_gather"])},
                "mpjpe_mm": ev["mpjpe_mm"]})
            # the whole pass as a rate - and the roofline of its dominant kernel on ONE full chunk of the first clip
            line["algorithmic_tflops"] = ev["algorithmic_tflops"]
            line["frac_of_fp32_mfma_peak"] = round(ev["algorithmic_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
            if world == 1 and mine and lifter.precision(dev) == "f32":
                def _clip_roofline():
                    nb = lifter.CLIP_CHUNK
                    c, padded, prow, _ = max(mine, key=lambda t: t[1].shape[0])
                    reps = -(-(nb + 242) // padded.shape[0])
                    clipb = torch.cat([padded] * reps, dim=0)[: nb + 242].contiguous()
                    run = lambda: lifter.forward_clip(clipb, prow)
                    with torch.no_grad():
                        run()
                        settle_clocks(run, dev, group=5, max_groups=20)
                        el_c, dev_c, _ = timed_steps(run, 20, 3, barrier, dev)
                        rl = roofline(lifter, clipb, prow, dev_c / 20 * 1e3, fn=run, batch=nb, key="eval_b%d" % nb)
                    rl.update({"windows": nb, "ms_per_call": round(el_c / 20 * 1e3, 4),
                               "note": "one clip call of %d windows (window stride one frame): the per-frame launch + r3d_forward_clip_f32 + "
                                       "decoder; algorithmic FLOPs count expand_conv per window, the kernel evaluates it per frame" % nb})
                    return rl
                try:
                    line["roofline"] = _clip_roofline()
                except Exception as e:                      # noqa: BLE001 - reported, not swallowed
                    line["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if world == 1 and args.lanes > 1 and lifter.precision(dev) == "f32":
                # SECONDARY: the same pass with the clips dealt to N library-owned lanes (R3D_OPT_LANES): one weight image, N clips in flight
                try:
                    lifter.set_lanes(args.lanes, dev)
                    ev2 = eval_pass(lifter, dev, dist, world, rank, args.clips, args.steps, args.warmup, barrier, max_over_ranks, all_ranks,
                                    lanes=args.lanes)
                    ev2.pop("_mine")
                    line["lanes"] = {"lanes": args.lanes, "value": ev2["value"], "unit": "poses/s", "ms_per_pass": ev2["ms_per_pass"],
                                     "mpjpe_mm": ev2["mpjpe_mm"],
                                     "mpjpe_abs_diff_mm": abs(ev2["mpjpe_mm"]["action_average"] - ev["mpjpe_mm"]["action_average"]),
                                     "checksum_rel_diff": abs(ev2["mpjpe_mm"]["checksum"] - ev["mpjpe_mm"]["checksum"]) / max(abs(ev["mpjpe_mm"]["checksum"]), 1e-30),
                                     "speedup_vs_one_lane": round(ev2["value"] / ev["value"], 4),
                                     "note": "SECONDARY: R3D_OPT_LANES on the pass's own handles; tile schedules of a lane are cut for its share of "
                                             "the CUs, so sums may differ from the whole-chip pass in the last bits (split-K pieces)"}
                    lifter.set_lanes(0, dev)
                except Exception as e:                      # noqa: BLE001 - reported, not swallowed
                    line["lanes"] = {"error": "%s: %s" % (type(e).__name__, e)}
            print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
