"""Build-time guard for the kernels whose pipelines COUNT vector-memory operations (s_waitcnt vmcnt(N) with N > 0 across
barriers: conv_stream, conv_fwd2, conv_wgrad2 / 2s / 3).  A register spill is a scratch load or store, i.e. a
vector-memory operation the counts do not know about -- the waits would silently cover the wrong loads.  hipcc's
-Rpass-analysis=kernel-resource-usage remarks (stderr of the conv.hip compile) are parsed here and the build fails if
one of those kernels uses scratch, or if a streaming kernel's compiler-allocated registers reach the ones the kernel
owns by hand (v208..v255, see conv.hip)."""
import re
import sys

COUNTED = ("conv_stream_kernel_t", "conv_fwd2_kernel", "conv_wgrad2_kernel", "conv_wgrad2s_kernel", "conv_wgrad3_kernel")


def main(path):
    cur, info = None, {}
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            info[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass-analysis", line)
        if m and cur:
            info[cur][m.group(1).strip()] = m.group(2)
    bad, seen = [], 0
    for name, r in info.items():
        if not any(k in name for k in COUNTED):
            continue
        seen += 1
        if r.get("ScratchSize [bytes/lane]") != "0" or r.get("VGPRs Spill") != "0" or r.get("Dynamic Stack") != "False":
            bad.append(f"{name}: scratch {r.get('ScratchSize [bytes/lane]')} B/lane, {r.get('VGPRs Spill')} spilled VGPRs")
    if seen < 8:
        bad.append(f"only {seen} counted-wait kernels found in {path}: remarks missing?")
    if bad:
        sys.stderr.write("check_resources: kernels with counted vmcnt waits must not touch scratch:\n  " + "\n  ".join(bad) + "\n")
        return 1
    print(f"check_resources: {seen} counted-wait kernels, no scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
