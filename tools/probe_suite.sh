#!/bin/bash
# usage: probe_suite.sh binA binB ...  : runs the level shapes of the RF-243 plan at B=256 through each probe binary
cd "$(dirname "$0")/.."
SHAPES=("6 6912 256 768" "6 6912 256 256" "6 2304 256 768" "6 2304 256 256" "6 768 256 768" "6 768 256 256" "6 256 256 768" "6 256 256 256" "6 256 1024 1024" "5 256 1024 832" "5 256 256 1024")
for s in "${SHAPES[@]}"; do
  for b in "$@"; do
    printf "%-22s %-28s " "$s" "$b"
    timeout 120 $b $s 30 2>&1 | grep -E "best|spot" | tr "\n" " "; echo
  done
done
