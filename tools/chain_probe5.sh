#!/bin/bash
mkdir -p gpurun_out
{
for b in tools/chain_probe4_*.bin; do echo "== $b 8"; timeout 120 $b 8; done
} > gpurun_out/chain_probe.txt 2>&1
grep "^==\|GATHER\|max abs" gpurun_out/chain_probe.txt
