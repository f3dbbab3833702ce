cd /root/repo
mkdir -p gpurun_out/r03w
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -45 > gpurun_out/r03w/pytest_full.log
for B in 1 2 4 8 16 32; do bash tools/ab_libs2.sh $B 2>&1 | grep libray3d > gpurun_out/r03w/ab_$B.log; done
