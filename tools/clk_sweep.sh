#!/bin/bash
# live shader clock of the single-launch forward (r3d_last_clock) by window count
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 64 128 192 256 320 384 512 768 1024 2048; do python bench.py --batch $B --no-cpu-baseline --no-b1024 --no-bf16x3 --no-shipped-cfgs --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print($B, l['ms_per_step'], 'clk', r['clk_ghz'], 'frac', r.get('frac'), 'at clk', r.get('frac_at_clk'))"; done
