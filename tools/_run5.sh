cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --half-chip-streams --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --steps 200 --warmup 20 > gpurun_out/r05_half_chip.json 2> gpurun_out/r05_half_chip.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_half_chip.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('two_stream_variant'), indent=1))
PY
tail -5 gpurun_out/r05_half_chip.err
