#!/bin/bash
# HBM traffic per kernel launch of the bench step: two separate counter passes (FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md's HBM section prescribes, over eager launches of the bench step.
# usage: scripts/pmc_traffic.sh TAG   -> gpurun_out/TAG_fetch, gpurun_out/TAG_write, gpurun_out/TAG_traffic.json
set -o pipefail
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_$n -o r -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_$n.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_$n.log; exit 1; }
done
cd $R && python scripts/pmc_traffic.py $TAG
