#!/bin/bash
# Same-box A/B of two ENVIRONMENT settings on the conv micro-benchmark (one library, a switch read at run time):
#   scripts/ab_env.sh "VAR=a" "VAR=b" KIND LAYER [LAYER ...]     order A B B A per layer; prints us per launch
set -o pipefail
EA=$1; EB=$2; KIND=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for layer in "$@"; do
  line="$layer $KIND"
  for which in A B B A; do
    e=$EA; [ $which = B ] && e=$EB
    us=$(env $e timeout -k 10 120 python scripts/bench_conv.py --kind $KIND --only $layer --iters 30 2>/dev/null | awk '{print $3}') || exit 1
    line="$line  $which=$us"
  done
  echo "$line"
done
