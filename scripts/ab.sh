#!/bin/bash
# Same-box A/B of two builds of libunet_hip.so on the headline step (clocks differ by ~12 % between boxes and drift
# by ~1 % within a run, so only an interleaved comparison on one box means anything):
#   scripts/ab.sh PATH_TO_LIB_A PATH_TO_LIB_B [extra bench.py args]
# Both libraries must live inside the tree (gpurun uploads the tree; gpurun_out/ is not uploaded).
# Order A B B A A B B A; prints ms/step of every run and the two means.
set -o pipefail
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for which in A B B A A B B A; do
  lib=$A; [ $which = B ] && lib=$B
  UNET_HIP_LIB=$lib timeout -k 10 150 python bench.py --no-cpu-baseline --no-kernel-timing --steps 10 "$@" > gpurun_out/ab_run.log 2> gpurun_out/ab_run.err || { tail -5 gpurun_out/ab_run.err; exit 1; }
  ms=$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_run.log | head -1 | cut -d' ' -f2)
  echo "$which $ms" | tee -a gpurun_out/ab.txt
done
awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%s mean %.3f ms over %d runs\n", k, s[k]/n[k], n[k]}' gpurun_out/ab.txt
