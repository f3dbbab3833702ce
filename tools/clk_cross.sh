cd $GRAFT_REPO_ROOT
one() { python bench.py --batch $1 --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('bench', $1, l['ms_per_step'], 'clk', r['clk_ghz'])"; }
one 256; python tools/clk_repeat.py 256 20 2>/dev/null | tail -1; one 256; one 1024; python tools/clk_repeat.py 1024 20 2>/dev/null | tail -1
