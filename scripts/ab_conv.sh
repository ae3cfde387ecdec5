#!/bin/bash
# Same-box A/B of two builds of libunet_hip.so on the conv micro-benchmark (boxes differ by up to ~12 % in clock, so only
# an interleaved comparison on one box means anything):
#   scripts/ab_conv.sh LIB_A LIB_B KIND LAYER [LAYER ...]      order A B B A per layer; prints us per launch
set -o pipefail
A=$1; B=$2; KIND=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for layer in "$@"; do
  line="$layer $KIND"
  for which in A B B A; do
    lib=$A; [ $which = B ] && lib=$B
    us=$(UNET_HIP_LIB=$lib timeout -k 10 120 python scripts/bench_conv.py --kind $KIND --only $layer --iters 30 2>/dev/null | awk '{print $3}') || exit 1
    line="$line  $which=$us"
  done
  echo "$line"
done
