cd ${GRAFT_REPO_ROOT:-/root/repo}
# An M = B launch (7 x 256 x 1024 x K) - cycles and shader clock per workgroup ("eff GHz": cycle counter against the 100 MHz wall
# clock) - launched with a host synchronisation after every launch and back to back (PROBE_NOSYNC), on random and on all-zero
# operands.
for sync in "" "PROBE_NOSYNC=1"; do for z in "" "PROBE_ZERO=1"; do for K in 1024 4096; do
  echo "== [$sync $z] K=$K"; env $sync $z ./tools/gemm_probe_timing.bin 7 256 1024 $K 200 | grep -E "best|back to back|wall span| [0-9]+: 1 " | head -5
done; done; done
