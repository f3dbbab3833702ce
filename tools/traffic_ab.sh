#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and step time of the 256-window forward under
# development switches of the hooks build: bash tools/traffic_ab.sh "<VAR=val ...>" "<VAR=val ...>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B=${TRAFFIC_BATCH:-256}
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  O=$R/gpurun_out/traffic_$tag
  rm -rf $O; mkdir -p $O
  ARGS="--batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024"
  env R3D_USE_HOOKS_LIB=1 $cfg rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -- python $R/bench.py $ARGS > $O/p3.log 2>&1
  env R3D_USE_HOOKS_LIB=1 $cfg rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p4 -- python $R/bench.py $ARGS > $O/p4.log 2>&1
  python - "$O" "$cfg" <<'PY'
import csv, glob, sys, os
root, cfg = sys.argv[1], sys.argv[2]
def last(d, ctr):
    f = max(glob.glob(d + '/**/*_counter_collection.csv', recursive=True), key=os.path.getmtime)
    out = {}
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].split('.')[0]
        if n.startswith('r3d_forward') or n.startswith('r3d_gemm'):
            out.setdefault((n, int(r['Dispatch_Id'])), {})[r['Counter_Name']] = float(r['Counter_Value'])
    k = max(out, key=lambda t: t[1])
    return k[0], out[k]
n3, c3 = last(root + '/p3', 'FETCH_SIZE'); n4, c4 = last(root + '/p4', 'WRITE_SIZE')
hit = 100 * c4['TCC_HIT_sum'] / max(1, c4['TCC_HIT_sum'] + c4['TCC_MISS_sum'])
print('[%s] %s: fetch %.1f MB  write %.1f MB  total %.1f MB  L2 hit %.1f %%' % (cfg, n3, c3['FETCH_SIZE'] * 2 / 1e3, c4['WRITE_SIZE'] / 1e3, c3['FETCH_SIZE'] * 2 / 1e3 + c4['WRITE_SIZE'] / 1e3, hit))
PY
  cd $R && bash tools/ab_env.sh $B "$cfg" && cd /tmp
done
