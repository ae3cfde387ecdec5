"""Build-time guard for the kernels whose pipelines COUNT vector-memory operations (s_waitcnt vmcnt(N) with N > 0 across
barriers: conv_stream, conv_fwd2, conv_wgrad2 / 2s / 3).  A register spill is a scratch load or store, i.e. a
vector-memory operation the counts do not know about -- the waits would silently cover the wrong loads.  hipcc's
-Rpass-analysis=kernel-resource-usage remarks (stderr of the conv.hip compile) are parsed here and the build fails if
one of those kernels uses scratch.  With a second argument (conv.o) the device code is disassembled and the streaming
kernels are scanned for any use of the registers they own by hand (v208..v255, see conv.hip) outside the hand-written
loads and copies."""
import glob
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OWNED_FIRST = 208        # conv.hip: ST_OWNED_FIRST

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


def owned_registers(obj):
    """Disassemble the device code of ``obj`` and check, for the streaming kernels, that the registers the kernel owns
    by hand (v208..v255) appear ONLY in the hand-written instructions: the buffer loads that fill them and the v_mov
    copies that read them.  Anything else means the compiler allocated one of them."""
    if not os.path.exists(OBJDUMP):
        print("check_resources: no llvm-objdump, owned-register scan skipped")
        return 0
    subprocess.run([OBJDUMP, "--offloading", obj], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    parts = glob.glob(obj + ".0.*")
    dev = [f for f in parts if "amdgcn" in f]
    try:
        if not dev:
            sys.stderr.write("check_resources: no device code found in " + obj + "\n")
            return 1
        dis = subprocess.run([OBJDUMP, "-d", dev[0]], check=True, capture_output=True, text=True).stdout
    finally:
        for f in parts:
            os.remove(f)
    cur, bad, used = None, [], {}
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if not cur or "conv_stream_kernel_t" not in cur:
            continue
        body = line.split("//")[0].rstrip()
        if "scratch_" in body or "v_accvgpr" in body:
            bad.append(f"{cur}: {body.strip()}")
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", body)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", body)]
        if any(r >= OWNED_FIRST for r in regs):
            ok = (re.search(r"buffer_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\], 0 offen", body)
                  or re.search(r"v_mov_b32_e32 v\d+, v\d+$", body))
            used[cur] = used.get(cur, 0) + 1
            if not ok:
                bad.append(f"{cur}: {body.strip()}")
    if bad:
        sys.stderr.write("check_resources: the streaming kernels' hand-owned registers are touched by other code:\n  "
                         + "\n  ".join(bad[:20]) + "\n")
        return 1
    if len(used) < 3:
        sys.stderr.write(f"check_resources: expected 3 streaming kernels that use owned registers, found {len(used)}\n")
        return 1
    print(f"check_resources: owned registers v{OWNED_FIRST}+ appear only in hand-written instructions ({sorted(used.values())})")
    return 0


if __name__ == "__main__":
    rc = main(sys.argv[1])
    if rc == 0 and len(sys.argv) > 2:
        rc = owned_registers(sys.argv[2])
    sys.exit(rc)
