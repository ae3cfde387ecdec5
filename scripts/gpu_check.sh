#!/bin/bash
# Round check on the GPU box: parity tests, default bench line, rocprofv3 kernel stats of the bench.
# usage: scripts/gpu_check.sh TAG   (outputs under gpurun_out/TAG_*)
set -o pipefail
TAG=${1:-chk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -3 gpurun_out/${TAG}_tests.log
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err || { tail -30 gpurun_out/${TAG}_bench.err; exit 1; }
cat gpurun_out/${TAG}_bench.log
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_prof.log 2>&1 || { tail -30 $R/gpurun_out/${TAG}_prof.log; exit 1; }
echo done
