#!/bin/bash
# counter passes over the band-pair attention kernels (forward + backward once at the config-B bottleneck shape):
#   scripts/pmc_attn.sh TAG      -> gpurun_out/TAG_{sq1,sq2}/r_counter_collection.csv
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$n -o r -- python3 $R/scripts/check_attn.py --time-only > $R/gpurun_out/${TAG}_$n.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_$n.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT && \
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES
echo pmc-done
# HBM traffic (separate passes; FETCH_SIZE / WRITE_SIZE in the guide's units and corrections) and L2 hit rate
if [ "$2" = "mem" ]; then
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && run fetch FETCH_SIZE && run write WRITE_SIZE
  echo pmc-mem-done
fi
