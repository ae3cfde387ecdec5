"""Diagnostic (one GPU): capture of the data-parallel step under torch's DEFAULT ("global") stream-capture mode while a
process group -- and with it the group's watchdog thread -- is alive.  DESIGN section 5 states why
graphs.GraphedTrainStep captures with "thread_local" in that situation; this script produces the evidence: it runs the
body of tests/test_gpu_dp_graph.py in a child process with UNET_CAPTURE_MODE=global, up to ``--tries`` times, stops at
the first child that fails, and writes each child's return code and the tail of its stderr to
gpurun_out/capture_mode_global.txt.  A control run under thread_local follows only if every global run succeeded."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_dp_graph import CHILD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tries", type=int, default=3)
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "capture_mode_global.txt"), "w")
    failed = False
    for i in range(args.tries):
        env = dict(os.environ, UNET_CAPTURE_MODE="global", TORCH_NCCL_ASYNC_ERROR_HANDLING="1")
        out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=150,
                             cwd=ROOT, env=env)
        ok = out.returncode == 0 and "DP-GRAPH-OK" in out.stdout
        log.write(f"=== global-mode capture, try {i}: rc={out.returncode} ok={ok}\n")
        log.write("--- stdout tail\n" + out.stdout[-1500:] + "\n--- stderr tail\n" + out.stderr[-6000:] + "\n")
        log.flush()
        print(f"global try {i}: rc={out.returncode} ok={ok}", flush=True)
        if not ok:
            failed = True
            break              # one failure is the evidence; nothing more is started on the GPU after an abort
    if not failed:
        out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=150,
                             cwd=ROOT, env=dict(os.environ, UNET_CAPTURE_MODE="thread_local"))
        log.write(f"=== thread_local control: rc={out.returncode}\n")
        print(f"thread_local control: rc={out.returncode}", flush=True)
    log.close()


if __name__ == "__main__":
    main()
