#!/bin/bash
# counter passes over the streaming 1x1 kernel's epilogue variants (scripts/bench_stream_epilogue.py):
#   scripts/pmc_stream.sh TAG   -> gpurun_out/TAG_{sq1,sq2,fetch,write}
set -o pipefail
TAG=${1:-pst}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$n -o r -- python $R/scripts/bench_stream_epilogue.py > $R/gpurun_out/${TAG}_$n.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_$n.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT && \
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES && \
run fetch FETCH_SIZE && run write WRITE_SIZE
echo pmc-done
