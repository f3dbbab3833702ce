cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "every_mode_and_form or staged or half_chip") > gpurun_out/r05_run7_tests.log 2>&1
grep -n "passed\|failed\|FAILED" gpurun_out/r05_run7_tests.log | tail -3
RND=r05 bash tools/final_measure.sh
python -c "
import json
for f in ('r05_bench_default','r05_bench_driver_cmd','r05_bench_eval'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"
cat gpurun_out/r05_batch_sweep.txt
