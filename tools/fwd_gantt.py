#!/usr/bin/env python3
"""Per-tile timeline of the single-launch forward (timing build of the library).

    bash tools/build_probe.sh                       (builds tools/libray3d_hip_timing.so)
    R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=all R3D_TIMING_DUMP=gpurun_out/gantt_256.txt \
        python tools/stage_times.py 256 1           (on the GPU box)
    python tools/fwd_gantt.py gpurun_out/gantt_256.txt

Reads the dump (P lines: problems; T lines: workgroup, tile, problem, units, row, column, split-K, producers, fetched /
ready / finished in us, tiles in the run) and prints: the launch's span, how long workgroups waited for producers, the
measured duration of a tile per (problem class, units, split-K), and the idle time per phase of the launch."""
import sys
from collections import defaultdict
probs, tiles = {}, []
for line in open(sys.argv[1]):
    f = line.split()
    if f[0] == "P":
        probs[int(f[1])] = dict(name=f[2], rpw=int(f[4]), M=int(f[6]), N=int(f[8]), K=int(f[10]), fused=int(f[12]))
    elif f[0] == "T":
        tiles.append(dict(wg=int(f[1]), t=int(f[2]), p=int(f[3]), mi=int(f[4]), row=int(f[5]), col=int(f[6]), ks=int(f[7]), ndep=int(f[8]),
                          fetch=float(f[9]), ready=float(f[10]), end=float(f[11]), run=int(f[12])))
stamped = [t for t in tiles if t["end"] > 0]
span = max(t["end"] for t in stamped)
print("tiles %d (stamped %d), span %.1f us, workgroups %d" % (len(tiles), len(stamped), span, len(set(t["wg"] for t in tiles))))
wait = sum(max(t["ready"] - t["fetch"], 0) for t in stamped if t["ready"] > 0)
nwg = len(set(t["wg"] for t in tiles))
print("waiting for producers: %.1f us per workgroup on average (%.1f %% of the span)" % (wait / nwg, 100 * wait / nwg / span))
last_end = defaultdict(float)
for t in stamped:
    last_end[t["wg"]] = max(last_end[t["wg"]], t["end"])
tail = sum(span - e for e in last_end.values()) / nwg
print("idle behind a workgroup's last tile: %.1f us on average (%.1f %%)" % (tail, 100 * tail / span))
def cls(name):
    for k in ("expand_conv", "layers_conv", "shrink", "embedder", "GlobalInfo", "FuseBlocks", "Integration"):
        if k in name:
            return ("trj." if name.startswith("LocalLayer.") or name.startswith("Integration.") else "") + k
    return name
dur = defaultdict(list)
for t in stamped:
    p = probs[t["p"]]
    key = (cls(p["name"]), p["fused"], p["K"], t["mi"], t["ks"], min(p["N"], 256))
    start = t["ready"] if t["ready"] > 0 else t["fetch"]
    dur[key].append((t["end"] - start) / max(t["run"], 1))
print("%-26s fused K mi ks N   n   median us  min  max" % "class")
for k in sorted(dur):
    v = sorted(dur[k])
    print("%-26s %d %5d %d %d %3d %4d  %7.2f %6.2f %6.2f" % (k[0], k[1], k[2], k[3], k[4], k[5], len(v), v[len(v) // 2], v[0], v[-1]))
# utilisation over time: busy workgroups in 10-us bins
bins = int(span // 10) + 1
busy = [0.0] * bins
for t in stamped:
    s0 = t["ready"] if t["ready"] > 0 else t["fetch"]
    b0, b1 = int(s0 // 10), int(t["end"] // 10)
    for b in range(b0, min(b1, bins - 1) + 1):
        lo, hi = max(s0, b * 10), min(t["end"], (b + 1) * 10)
        if hi > lo:
            busy[b] += hi - lo
print("busy workgroups per 10-us bin:", " ".join("%d" % round(v / 10) for v in busy))
