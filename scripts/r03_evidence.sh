#!/bin/bash
# Round-3 evidence run on the GPU box: counter passes for the kernel variant added this round (256-position tile of
# conv_fwd2 on the bottleneck conv, against the 128-position tile), the other workloads of bench.py, the inference
# benchmark.  Outputs under gpurun_out/r3ev_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
UNET_CONV_BN=128 bash scripts/pmc_conv.sh r3ev_mid128 mid fwd || exit 1
bash scripts/pmc_conv.sh r3ev_mid256 mid fwd || exit 1
bash scripts/pmc_conv.sh r3ev_dec0wg dec0_m wgrad || exit 1
python scripts/bench_conv.py > gpurun_out/r3ev_conv_all.txt 2>/dev/null || exit 1
for w in B2 C; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r3ev_bench_$w.json 2> gpurun_out/r3ev_bench_$w.err || exit 1
done
timeout -k 10 400 python bench.py --attention cross_freq_v2 --no-cpu-baseline > gpurun_out/r3ev_bench_cfv2.json 2> gpurun_out/r3ev_bench_cfv2.err || exit 1
timeout -k 10 400 python bench.py --conditioning spectro_temporal --no-cpu-baseline > gpurun_out/r3ev_bench_st.json 2> gpurun_out/r3ev_bench_st.err || exit 1
timeout -k 10 300 python scripts/bench_infer.py --model B > gpurun_out/r3ev_infer_B.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/bench_infer.py --model phase5 > gpurun_out/r3ev_infer_p5.txt 2>&1 || exit 1
echo evidence-done
