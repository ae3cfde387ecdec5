#!/bin/bash
# Same-box A/B of the whole bench step under two settings of one environment variable (boxes differ by a few % in clock):
#   scripts/ab_bench.sh VAR VALUE_A VALUE_B [bench.py args]      order A B B A; prints ms per step
set -o pipefail
VAR=$1; A=$2; B=$3; shift 3
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for which in A B B A; do
  val=$A; [ $which = B ] && val=$B
  ms=$(env $VAR=$val timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 "$@" 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))") || exit 1
  echo "$VAR=$val  $ms ms/step"
done
