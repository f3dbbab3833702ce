#!/bin/bash
# Why is the register-chained first-level tile slower inside the forward than alone?  
#  1. pos alone / trj alone / pair, chain against taps: does the penalty need the trajectory tiles in between?
#  2. instruction-cache and L2 counters of the forward kernel, chain against taps (one counter per pass; unknown names are skipped)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out/chain_why
O=$R/gpurun_out/chain_why
for rep in 1 2; do
  for c in 1 0; do
    for B in 256 1024; do R3D_USE_HOOKS_LIB=1 R3D_CHAIN=$c python tools/chain_pos_only.py $B 200 2>&1 | grep "^CHAIN"; done
  done
done | tee $O/pos_only_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "\b[A-Z_]*\(ICACHE\|IFETCH\|INST_CACHE\|SQC_INST\)[A-Za-z_]*" | sort -u > $O/icache_counters.txt
cat $O/icache_counters.txt
for ctr in $(cat $O/icache_counters.txt | head -12) SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum FETCH_SIZE TCP_TCC_READ_REQ_sum SQ_INSTS_SALU SQ_WAIT_INST_LDS; do
  for c in 1 0; do
    d=$O/pmc_${ctr}_c$c
    R3D_USE_HOOKS_LIB=1 R3D_CHAIN=$c timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- python $R/bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" $ctr $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("r3d_forward")]
by = {}
for r in rows:
    by.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, cn), v in sorted(by.items()):
    print("%-28s chain=%s %-28s last dispatch %.4g  (n %d, median %.4g)" % (cn, sys.argv[3], k, v[-1], len(v), sorted(v)[len(v) // 2]))
PY
    else
      echo "$ctr chain=$c: no counter file ($(tail -1 $d.log | cut -c1-120))"
    fi
    rm -rf $d
  done
done 2>&1 | tee $O/pmc_chain_vs_taps.txt
