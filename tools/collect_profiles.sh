#!/bin/bash
# gpurun_out/<round>_* (tools/final_measure.sh) -> profiles/: the committed summaries, then profiles/pmc.json
cd /root/repo
RND=${RND:-r06}
for d in gpurun_out/${RND}_*/; do
  n=$(basename $d)
  mkdir -p profiles/$n
  for f in kernel_stats.csv bench_line.json pmc_table.txt pmc_summary.json fwd_gantt.txt; do [ -f $d/$f ] && cp $d/$f profiles/$n/; done
done
for f in gpurun_out/${RND}_bench_*.json gpurun_out/${RND}_batch_sweep.txt; do [ -f $f ] && cp $f profiles/; done
python profiles/make_pmc_json.py $RND
