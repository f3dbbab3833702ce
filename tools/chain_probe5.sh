#!/bin/bash
mkdir -p gpurun_out
{
for n in 1 2 8; do for b in tools/chain_probe4_dma.bin tools/chain_probe4_dma_img5.bin; do echo "== $b $n"; timeout 120 $b $n; done; done
} > gpurun_out/chain_probe.txt 2>&1
grep "^==\|NIMG\|max abs" gpurun_out/chain_probe.txt
