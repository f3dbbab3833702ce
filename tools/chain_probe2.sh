#!/bin/bash
# runs the chained-tile probes on the GPU box (gpurun -- bash tools/chain_probe2.sh [pattern]); output -> gpurun_out/chain_probe.txt
mkdir -p gpurun_out
pat=${1:-tools/chain_probe[23]*.bin}
{
for b in $pat; do
  [ -x "$b" ] || continue
  echo "== $b"
  timeout 120 $b 8 || echo "FAILED rc=$?"
done
} > gpurun_out/chain_probe.txt 2>&1
cat gpurun_out/chain_probe.txt
