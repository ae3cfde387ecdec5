#!/bin/bash
# rocprofv3 kernel stats of the bench step under VAR=VALUE (exported before the profiler starts: no wrapper after --)
#   scripts/prof_env.sh TAG VAR VALUE
set -o pipefail
TAG=$1; export $2=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/$TAG.log 2>&1
