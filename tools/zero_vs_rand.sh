cd $GRAFT_REPO_ROOT
# the shader clock of an M = B launch (cycle counter against the 100 MHz wall clock per workgroup: "eff GHz"): random / zero
# operands; every problem streaming the same 4 MB of weights (L2-resident) instead of its own
for z in "" "PROBE_ZERO=1" "PROBE_SHARE_W=1" "PROBE_SHARE_W=1 PROBE_SHARE_A=1"; do for K in 1024 4096; do echo "== [$z] K=$K"; env $z ./tools/gemm_probe_timing.bin 7 256 1024 $K 30 | grep -E "best|wall span| [0-9]+: 1 " | head -5; done; done
