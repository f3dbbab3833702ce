cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16x3 or reference_fixture_in_every" 2>&1 | tail -5
python tools/bf16x3_error_table.py 6 128 2>/dev/null | tail -8
for B in 256 1024; do R3D_BF16X3=1 python bench.py --batch $B --no-cpu-baseline --no-shipped-cfgs --no-b1024 --no-bf16x3 --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', $B, l['ms_per_step'], l['value'], l['dtype'], l['roofline']['frac'], l['parity_max_abs_err'])"; done
