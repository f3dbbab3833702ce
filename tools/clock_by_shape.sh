cd $GRAFT_REPO_ROOT
# shader clock (cycles / wall clock per workgroup) of the eight-wave GEMM launch by tile height: M = 256 (32-row tiles), 512 (64), 1024 (128)
for M in 256 512 1024 1536; do echo "== M=$M"; ./tools/gemm_probe_timing.bin 7 $M 1024 1024 30 | grep -E "grid|best|wall span| [0-9]+: 1 " | head -6; done
echo "== pair-shaped: 6 problems M 6912 N 256 K 768"; ./tools/gemm_probe_timing.bin 6 6912 256 768 20 | grep -E "grid|best|wall span| [0-9]+: [0-9] " | head -6
