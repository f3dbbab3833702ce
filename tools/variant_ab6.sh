#!/bin/bash
# three calls of tools/variant_ab.sh (six rounds) for one set of variants, the first with the eval pass: bash tools/variant_ab6.sh "<names>"
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/variant_check.sh "$1" eval | grep "^==\|^base\|^$(echo $1 | cut -d' ' -f1)" > gpurun_out/variant_ab6.txt
for i in 1 2; do bash tools/variant_ab.sh "$1" | grep -v "^$" >> gpurun_out/variant_ab6.txt; done
cat gpurun_out/variant_ab6.txt
